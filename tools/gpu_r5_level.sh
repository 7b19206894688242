#!/bin/bash
# round 5, VERDICT item 6: the headline launch's level across processes, placements of the output arrays and runtime settings, one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5lvl}
mkdir -p $O
cd $R
export TMPDIR=/tmp
F=$O/r05_process_level.txt
echo "G env-steps/s, median (slowest..fastest) of 12 launches of 4 000 steps x 65 536 cramped_room envs; one line = one process" > $F
for i in 1 2 3 4 5; do timeout 120 python3 tools/process_level.py default_$i 2>/dev/null | tail -1 >> $F; done
HSA_ENABLE_SDMA=0 timeout 120 python3 tools/process_level.py HSA_ENABLE_SDMA=0 2>/dev/null | tail -1 >> $F
HSA_XNACK=1 timeout 120 python3 tools/process_level.py HSA_XNACK=1 2>/dev/null | tail -1 >> $F
GPU_MAX_HW_QUEUES=1 timeout 120 python3 tools/process_level.py GPU_MAX_HW_QUEUES=1 2>/dev/null | tail -1 >> $F
HIP_FORCE_DEV_KERNARG=0 timeout 120 python3 tools/process_level.py HIP_FORCE_DEV_KERNARG=0 2>/dev/null | tail -1 >> $F
HSA_CU_MASK=0:0-247 timeout 120 python3 tools/process_level.py "HSA_CU_MASK=0:0-247" 2>/dev/null | tail -1 >> $F
OC_AMD_LIB=$R/overcooked_ai_amd/noxcd.so timeout 120 python3 tools/process_level.py "no xcd_block (control)" 2>/dev/null | tail -1 >> $F
for i in 6 7; do timeout 120 python3 tools/process_level.py default_$i 2>/dev/null | tail -1 >> $F; done
cat $F
