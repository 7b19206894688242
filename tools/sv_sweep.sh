#!/bin/bash
# round 6: the wait / placement knobs of the resident batched step (oc_amd.hip sv_knobs) on a tuning build —
#   python tools/build_variants.py sv_tune=-DOC_AMD_TUNING ; gpurun -- 'bash tools/sv_sweep.sh'
# knob word: bits 0..7 naps before the first look, 8..15 naps between looks, bit 16 light polls, bit 17 no looks through the L2,
# bits 24..26 (client) XCD shift of its block claims; OC_SV_DEBUG=1 prints where a round trip goes.  Results: profiles/r06_step_server.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6sv
export OC_AMD_LIB=$PWD/overcooked_ai_amd/sv_tune.so
run() { echo "server=$1 client=$2 $3 $4"; OC_SV_SERVER=$1 OC_SV_CLIENT=$2 timeout 60 python tools/time_step_server.py ${3:-cramped_room} ${4:-65536} 1000 2>&1 | grep "resident\|oc_step_server\|  wg" | tail -${5:-2}; }
OC_SV_DEBUG=1 run 0x0100 0x1000100 cramped_room 65536
run 0x0100 0x1000100 cramped_room 65536
run 0x0100 0x2000100 cramped_room 65536
run 0x0100 0x4000100 cramped_room 65536
run 0x0100 0x7000100 cramped_room 65536
run 0x0100 0x4000100 cramped_room 16384
run 0x0100 0x4000100 cramped_room 256
run 0x0100 0x0000100 cramped_room 256
