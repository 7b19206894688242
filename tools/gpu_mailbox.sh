#!/bin/bash
# the single-env mailbox only: its GPU tests and where a step's time goes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-mb}
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_mailbox.py tests/test_gpu_dropin_api.py tests/test_gpu_live_reference.py -m gpu -q -x > $O/pytest_mailbox.log 2>&1; echo "pytest rc=$?" >> $O/pytest_mailbox.log; tail -4 $O/pytest_mailbox.log
timeout 120 python tools/time_single_env.py 2>&1 | grep -v amdgpu.ids | head -3 | tee $O/single_env.txt
