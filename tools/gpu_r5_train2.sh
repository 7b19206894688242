#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5train}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -x -q -m gpu -k "training_step_with_its" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
export OC_AMD_LIB=$R/overcooked_ai_amd/tuning.so
for g in 0 4 8 12; do
  OC_TRAIN_OBS_G=$g timeout 120 python3 tools/time_train_step.py cramped_room 65536 2>/dev/null | grep "use_phi=True" | sed "s/^/G=$g: /"
done
for g in 0 4; do
  OC_TRAIN_OBS_G=$g timeout 120 python3 tools/time_train_step.py asymmetric_advantages 65536 2>/dev/null | grep "use_phi=True" | sed "s/^/G=$g: /"
done
