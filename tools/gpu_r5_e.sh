#!/bin/bash
# round 5, evidence on the committed sources: gpu suite, smoke, the driver's exact command, rocprofv3 kernel trace of it
# (extras on: every side-leg kernel gets a duration row).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r05e}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r05_pytest_gpu.log; tail -4 $O/r05_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r05_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
python3 - <<PY
import json
d=json.load(open("$O/r05_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s traffic %s issue %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches"), d["roofline"].get("traffic"), d["roofline"].get("issue_bound")))
for k,v in (d.get("configs") or {}).items():
    print("config", k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("parity_check") or {}).get("mismatches"), v.get("error"))
print("single_env", d.get("single_env_api",{}).get("value"), "ref", d.get("cpu_baseline",{}).get("reference_python",{}).get("value"))
print("encode", {k:(v.get("frac") if isinstance(v,dict) else v) for k,v in d["encode"].items()})
print("training", d["training_env"])
print("step_api", d.get("step_api"))
PY
ls $O
