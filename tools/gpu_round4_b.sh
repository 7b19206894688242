#!/bin/bash
# round 4, GPU call B: gpu suite (mailbox tests first, under a short timeout of their own), single-env timing, driver command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r04b}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_mailbox.py -m gpu -q -x > $O/pytest_mailbox.log 2>&1; echo "mailbox pytest rc=$?" >> $O/pytest_mailbox.log; tail -15 $O/pytest_mailbox.log
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 120 python tools/time_single_env.py > $O/single_env.txt 2>&1; tail -12 $O/single_env.txt
OC_AMD_NO_MAILBOX=1 timeout 120 python tools/time_single_env.py > $O/single_env_nomailbox.txt 2>&1; tail -4 $O/single_env_nomailbox.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
python3 - <<PY
import json
d=json.load(open("$O/r04_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches")))
for k,v in (d.get("configs") or {}).items():
    print("config", k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("parity_check") or {}).get("mismatches"), v.get("error"))
print("single_env", d.get("single_env_api"))
PY
