#!/bin/bash
# round 4, GPU call A: gpu suite (new: sharded env, live reference, >32-layout one-step, urgency horizon, expected_0/1),
# the driver's exact command with the new step definition + side legs, rocprofv3 kernel trace of the same command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
timeout 300 python3 bench.py --gpus 2 --single-process --steps 2 --warmup 1 --envs 32768 > $O/r04_single_process_2shards_1gpu.json 2> $O/bench_sp.err; echo "single-process rc=$?"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r04_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
python3 - <<PY
import json
d=json.load(open("$O/r04_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s traffic %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches"), d["roofline"].get("traffic")))
for k,v in (d.get("configs") or {}).items():
    print("config", k, {kk:(vv if not isinstance(vv,dict) else {a:b for a,b in vv.items() if a in ("frac","mismatches")}) for kk,vv in v.items() if kk in ("value","launch_ms","roofline","parity_check","error","timed_region_s")})
print("single_env", d.get("single_env_api",{}).get("value"), "ref", d.get("cpu_baseline",{}).get("reference_python",{}).get("value"), d.get("cpu_baseline",{}).get("reference_python",{}).get("same_run"))
PY
ls $O
