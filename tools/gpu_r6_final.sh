#!/bin/bash
# round 6, the profiles of the final sources: gpu suite, smoke, the driver's exact command (+ rocprofv3 kernel trace of it), the
# single-process sharded form, 1 M envs, the nccl init path, single-env timing, SQ counters (headline + configs[3]), soak.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r06final}
mkdir -p $O
cd $R
export TMPDIR=/tmp
# SQ counters first: the bench line replays them as roofline.issue_bound (keyed to the source hash)
STEPS=4000 TILED8=1 timeout 600 bash tools/pmc_rollout.sh r06f > /dev/null 2>&1
cp gpurun_out/pmc_r06f.txt $O/r06_pmc_rollout.txt; cp gpurun_out/sq_counters_r06f.json $O/sq_counters.json; cp gpurun_out/sq_counters_r06f.json profiles/sq_counters.json
CONFIG=4 STEPS=1200 TILED8=1 timeout 600 bash tools/pmc_rollout.sh r06f4 > /dev/null 2>&1
cp gpurun_out/pmc_r06f4.txt $O/r06_pmc_rollout_config4.txt; cp gpurun_out/sq_counters_r06f4.json $O/r06_sq_counters_config4.json
timeout 1500 python -m pytest tests -m gpu -q > $O/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r06_pytest_gpu.log; tail -4 $O/r06_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
timeout 300 python3 bench.py --gpus 2 --single-process --steps 2 --warmup 1 --envs 32768 > $O/r06_single_process_2shards_1gpu.json 2> $O/bench_sp.err; echo "single-process rc=$?"
timeout 300 python3 bench.py --gpus 2 --backend gloo --share-device --envs 32768 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > $O/r06_two_ranks_one_gpu_gloo.json 2> $O/bench_gloo.err; echo "two ranks over gloo rc=$?"
timeout 300 python3 bench.py --envs 1048576 --steps 1 --warmup 1 --launches-per-step 20 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/r06_bench_1M_envs.json 2>> $O/bench_other.err
OC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > $O/r06_force_dist_nccl_1rank.json 2> $O/r06_force_dist_nccl_1rank.err; echo "nccl rc=$?"
for c in 3 4 5; do timeout 400 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/r06_bench_config$c.json 2>> $O/bench_other.err; done
timeout 120 python tools/time_single_env.py 2>&1 | grep -v amdgpu.ids | head -3 > $O/r06_single_env_final.txt; cat $O/r06_single_env_final.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r06_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
timeout 420 python tools/soak.py --seeds 40 --envs 4096 --steps 500 > $O/r06_soak.log 2>&1; echo "soak rc=$?" >> $O/r06_soak.log; tail -3 $O/r06_soak.log
python3 - <<PY
import json
d=json.load(open("$O/r06_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s traffic %s issue %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches"), d["roofline"].get("traffic"), d["roofline"].get("issue_bound")))
print("step_layout", d["roofline"].get("step_layout"))
for k,v in (d.get("configs") or {}).items():
    print("config", k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("parity_check") or {}).get("mismatches"), v.get("error"))
print("single_env", d.get("single_env_api",{}).get("value"), "ref", d.get("cpu_baseline",{}).get("reference_python",{}).get("value"))
print("encode", {k:(v.get("frac") if isinstance(v,dict) else v) for k,v in d["encode"].items()})
print("training", d["training_env"])
for f in ("r06_bench_1M_envs.json","r06_single_process_2shards_1gpu.json","r06_force_dist_nccl_1rank.json","r06_bench_config3.json","r06_bench_config4.json","r06_bench_config5.json"):
    try:
        j=json.load(open("$O/"+f)); print(f, "%.4g G" % (j["value"]/1e9), (j.get("roofline") or {}).get("frac"), (j.get("roofline") or {}).get("traffic"), (j.get("parity_check") or {}).get("mismatches"))
    except Exception as e: print(f, "ERR", e)
PY
ls $O
