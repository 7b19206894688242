#!/bin/bash
# round 5: k_encode_window against the env-group kernels on one box (tuning build: OC_ENC_WINDOW=0 = the old kernels)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5enc}
mkdir -p $O
cd $R
export TMPDIR=/tmp
export OC_AMD_LIB=$R/overcooked_ai_amd/tuning.so
for lay in ${LAYOUTS:-asymmetric_advantages}; do
  for cfg in ${CFGS:-0:256:0 16384:256:0 16384:256:1 32768:256:0 32768:256:1 32768:512:1 65536:256:1 65536:512:1 65536:1024:1 131072:512:1}; do
    IFS=: read w t x <<< "$cfg"
    if [ "$x" = "1" ]; then export OC_ENC_XCD=1; else unset OC_ENC_XCD; fi
    OC_ENC_WINDOW=$w OC_ENC_THREADS=$t timeout 120 python3 tools/time_encode.py $lay ${ENVS:-65536} 2>/dev/null | sed "s/^/win $w thr $t xcd $x: /" | tee -a $O/r05_encode_window.txt
  done
done
