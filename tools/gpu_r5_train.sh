#!/bin/bash
# round 5: k_train_step_obs (the training step and its observation in one kernel) — parity, then timing against the two-kernel path
# (tuning build: OC_TRAIN_NO_FUSED_OBS=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r5train}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -x -q -m gpu -k "${TESTK:-training_step or multi_agent or fused_training}" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
export OC_AMD_LIB=$R/overcooked_ai_amd/tuning.so
for lay in cramped_room asymmetric_advantages; do
  timeout 120 python3 tools/time_train_step.py $lay 65536 2>/dev/null | sed "s/^/one kernel : /" | tee -a $O/r05_train_step_obs.txt
  OC_TRAIN_NO_FUSED_OBS=1 timeout 120 python3 tools/time_train_step.py $lay 65536 2>/dev/null | sed "s/^/two kernels: /" | tee -a $O/r05_train_step_obs.txt
done
