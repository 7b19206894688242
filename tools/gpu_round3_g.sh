#!/bin/bash
# round 3, GPU call G (the frozen build after k_step1 / k_train_step1): gpu suite, the driver's exact command (+ rocprofv3 trace and PMC passes of it), SQ
# counters of the headline and the per-env-terrain kernels, the other BASELINE configs, 1-rank RCCL path, soak.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r03g}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_driver_cmd_bench.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
for c in 3 4 5; do
  timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 > $O/r03_bench_config$c.json 2>> $O/bench_other.err; echo "config $c rc=$?"
done
timeout 300 python3 bench.py --config 5 --envs 131072 --steps 4000 --warmup 400 --no-cpu-baseline > $O/r03_bench_config5_131072.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --envs 1048576 --steps 4000 --warmup 400 --no-extras --no-cpu-baseline --no-traffic --no-parity-check > $O/r03_bench_1M_envs.json 2>> $O/bench_other.err
OC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-traffic > $O/r03_force_dist_nccl_1rank.json 2> $O/r03_force_dist_nccl_1rank.err; echo "nccl rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-traffic > $O/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity-check --no-traffic > $O/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity-check --no-traffic > $O/bench_write.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r03_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
cd $R
STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r03g > /dev/null 2>&1
cp gpurun_out/pmc_r03g.txt $O/r03_pmc_rollout.txt; cp gpurun_out/sq_counters_r03g.json $O/sq_counters.json
CONFIG=5 STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r03g_c5 > /dev/null 2>&1
cp gpurun_out/pmc_r03g_c5.txt $O/r03_pmc_rollout_config5.txt; cp gpurun_out/sq_counters_r03g_c5.json $O/r03_sq_counters_config5.json
CONFIG=4 STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r03g_c4 > /dev/null 2>&1
cp gpurun_out/pmc_r03g_c4.txt $O/r03_pmc_rollout_config4.txt; cp gpurun_out/sq_counters_r03g_c4.json $O/r03_sq_counters_config4.json
# the one-step API: a C loop over oc_step (k_step1 vs k_step3), kernel durations under rocprofv3, the Python wrapper's pieces
{
  [ -x gpurun_scratch/step_loop ] || hipcc -O2 -w -o gpurun_scratch/step_loop tools/step_loop.cpp -ldl
  python3 - <<'PY'
from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
for nm in ("cramped_room", "asymmetric_advantages"):
    open("gpurun_scratch/%s.bin" % nm, "wb").write(LayoutTable([spec_from_name(nm)]).records.tobytes())
PY
  for lay in cramped_room asymmetric_advantages; do
    echo "== $lay, oc_step in place (k_step1)"; ./gpurun_scratch/step_loop overcooked_ai_amd/liboc_amd.so gpurun_scratch/$lay.bin 65536 20000 2>&1 | grep -v amdgpu.ids
    echo "== $lay, OC_STEP_NO_LEAN=1 (k_step3)"; OC_STEP_NO_LEAN=1 ./gpurun_scratch/step_loop overcooked_ai_amd/liboc_amd.so gpurun_scratch/$lay.bin 65536 20000 2>&1 | grep -v amdgpu.ids
  done
  echo "== Python wrapper"; python3 tools/time_step_host.py 2>&1 | grep -v amdgpu.ids
  echo "== training step"; python3 tools/time_train_step.py cramped_room 65536 2>&1 | grep "per step"
  cd /tmp
  for v in lean table; do
    rm -rf /tmp/p_$v
    if [ $v = table ]; then export OC_STEP_NO_LEAN=1; else unset OC_STEP_NO_LEAN; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o t -- python3 $R/tools/time_step.py cramped_room 65536 > /tmp/p_$v.log 2>&1
    echo "== rocprofv3 kernel durations, tools/time_step.py cramped_room 65536 ($v)"
    python3 - "$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:2]:
    print("   %-70s calls %s avg %.2f us min %.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
  unset OC_STEP_NO_LEAN
  cd $R
} > $O/r03_one_step.txt 2>&1
timeout 420 python tools/soak.py --seeds 48 --envs 4096 --steps 500 > $O/r03_soak.log 2>&1; echo "soak rc=$?" >> $O/r03_soak.log; tail -3 $O/r03_soak.log
for f in $O/r03_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print('%.1f G env-steps/s frac %.3f launch_ms %.4f parity %s traffic %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['launch_ms'], (d.get('parity_check') or {}).get('mismatches'), d['roofline'].get('traffic')))" 2>&1 | tail -1)"; done
ls $O
