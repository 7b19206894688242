#!/bin/bash
# the lean last pass of round 4 (GPU minutes nearly spent): gpu suite (stops at the first failure), the driver's command, one
# control run with the [step][env] flags layout, the rocprofv3 trace of the driver's command, the SQ counter passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r04l}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 1; fi
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check --flags-layout step > $O/control_step_layout.json 2>> $O/bench_other.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r04_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
TILED8=1 STEPS=4000 timeout 400 bash tools/pmc_rollout.sh r04l > /dev/null 2>&1
cp gpurun_out/pmc_r04l.txt $O/r04_pmc_rollout.txt; cp gpurun_out/sq_counters_r04l.json $O/sq_counters.json
python3 - <<PY
import json
d=json.load(open("$O/r04_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s traffic %s layout %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches"), d["roofline"].get("traffic"), d["config"].get("flags_layout","")[:22]))
print("store_only", (d["roofline"].get("store_only") or {}).get("env_steps_per_s"), (d["roofline"].get("store_only") or {}).get("rollout_over_store_only"))
for k,v in (d.get("configs") or {}).items():
    print("config", k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("parity_check") or {}).get("mismatches"), v.get("error"))
c=json.load(open("$O/control_step_layout.json"))
print("control [step][env] layout: %.1f G frac %.3f launch_ms %.4f layout %s" % (c["value"]/1e9, c["roofline"]["frac"], c["roofline"]["launch_ms"], c["config"].get("flags_layout")))
PY
