#!/bin/bash
# round 4, the profiles of the final sources: gpu suite, the driver's exact command (+ rocprofv3 kernel trace of it), the
# single-process sharded form, SQ counters of the headline kernel, where a single-env step's time goes, soak.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r04f}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_driver_cmd_bench.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd.time
if [ -n "${CONTROL_LIB:-}" ]; then  # another build's headline on this box, same sustained region (it must export what _lib.py binds)
  for rep in 1 2; do OC_AMD_LIB=$R/$CONTROL_LIB timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/control_long_$rep.json 2>> $O/bench_other.err; done
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/shipped_long_2.json 2>> $O/bench_other.err
fi
timeout 300 python3 bench.py --gpus 2 --single-process --steps 2 --warmup 1 --envs 32768 > $O/r04_single_process_2shards_1gpu.json 2> $O/bench_sp.err; echo "single-process rc=$?"
timeout 300 python3 bench.py --envs 1048576 --steps 1 --warmup 1 --launches-per-step 20 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/r04_bench_1M_envs.json 2>> $O/bench_other.err
OC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > $O/r04_force_dist_nccl_1rank.json 2> $O/r04_force_dist_nccl_1rank.err; echo "nccl rc=$?"
timeout 120 python tools/time_single_env.py 2>&1 | grep -v amdgpu.ids | head -3 > $O/r04_single_env.txt; cat $O/r04_single_env.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r04_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r04f > /dev/null 2>&1
cp gpurun_out/pmc_r04f.txt $O/r04_pmc_rollout.txt; cp gpurun_out/sq_counters_r04f.json $O/sq_counters.json
timeout 420 python tools/soak.py --seeds 40 --envs 4096 --steps 500 > $O/r04_soak.log 2>&1; echo "soak rc=$?" >> $O/r04_soak.log; tail -3 $O/r04_soak.log
python3 - <<PY
import json
d=json.load(open("$O/r04_driver_cmd_bench.json"))
print("headline %.1f G frac %.3f launch_ms %.4f region %.2fs parity %s traffic %s" % (d["value"]/1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["timed_region_s"], (d.get("parity_check") or {}).get("mismatches"), d["roofline"].get("traffic")))
for k,v in (d.get("configs") or {}).items():
    print("config", k, v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("parity_check") or {}).get("mismatches"), v.get("error"))
print("single_env", d.get("single_env_api",{}).get("value"), "ref", d.get("cpu_baseline",{}).get("reference_python",{}).get("value"))
print("featurize", d["encode"]["featurize_state"]["frac"], "training", d["training_env"]["obs_u8"]["us_per_batched_step"])
import glob
for f in sorted(glob.glob("$O/control_long_*.json") + glob.glob("$O/shipped_long_*.json")):
    j=json.load(open(f)); print(f.split("/")[-1], "%.1f G" % (j["value"]/1e9), j["roofline"]["frac"], j["roofline"]["launch_ms"])
for f in ("r04_bench_1M_envs.json","r04_single_process_2shards_1gpu.json","r04_force_dist_nccl_1rank.json"):
    try:
        j=json.load(open("$O/"+f)); print(f, "%.1f G" % (j["value"]/1e9), (j.get("roofline") or {}).get("frac"), (j.get("parity_check") or {}).get("mismatches"))
    except Exception as e: print(f, "ERR", e)
PY
ls $O
