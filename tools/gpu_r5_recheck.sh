#!/bin/bash
# round 5: the gpu suite, smoke and the driver's exact command once more on the last commit (python-side changes behind the final profiles)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r05recheck}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r05_pytest_gpu.log; tail -3 $O/r05_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-200
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
python3 - <<PY
import json
d=json.load(open("$O/r05_driver_cmd_bench.json"))
r=d["roofline"]
print("headline %.1f G frac %.3f parity %s traffic %.4f x issue %s step_layout %.1f G" % (d["value"]/1e9, r["frac"], d["parity_check"]["mismatches"], r["traffic"]/r["bytes_per_launch"], "present" if r.get("issue_bound") else None, r["step_layout"]["env_steps_per_s"]/1e9))
for k,v in d["configs"].items(): print("config", k, "%.4g G" % (v["value"]/1e9), "%.3f" % v["roofline"]["frac"], v["roofline"].get("traffic"), v["parity_check"]["mismatches"])
print("single_env %.0f ref %.0f" % (d["single_env_api"]["value"], d["cpu_baseline"]["reference_python"]["value"]))
print("training", d["training_env"]["obs_u8"]["us_per_batched_step"], d["training_env"]["obs_f32"]["us_per_batched_step"])
print("encode", d["encode"]["u8"]["frac"], d["encode"]["f32"]["frac"], d["encode"]["featurize_state"]["frac"])
PY
