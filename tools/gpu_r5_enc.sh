#!/bin/bash
# round 5: k_encode_waves against the workgroup-level encode kernels on one box (tuning build: OC_ENC_WAVES=0 = the old kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5enc}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -x -q -m gpu -k "encod or smoke or ragged or multi_agent" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
export OC_AMD_LIB=$R/overcooked_ai_amd/tuning.so
for lay in ${LAYOUTS:-asymmetric_advantages cramped_room}; do
  for cfg in ${CFGS:-0:0 8:0 4:0 8:4 4:4 4:8 16:0}; do
    IFS=: read w g <<< "$cfg"
    OC_ENC_WAVES=$w OC_ENC_G=$g timeout 120 python3 tools/time_encode.py $lay ${ENVS:-65536} 2>/dev/null | sed "s/^/waves $w G $g: /" | tee -a $O/r05_encode_waves.txt
  done
done
