#!/bin/bash
# round 3, GPU call B: full gpu suite on the CW = 4 / blocked-rows build, A/B of the headline launch against the round-2
# library on the SAME box, SQ + instruction-cache counters of the headline kernel, the other configs again.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r03b}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
Q="--no-extras --no-cpu-baseline --no-parity-check --no-traffic"
for rep in 1 2; do
  [ -f gpurun_scratch/liboc_r2.so ] && OC_AMD_LIB=$R/gpurun_scratch/liboc_r2.so timeout 200 python3 bench.py $Q > $O/ab_r2_$rep.json 2>> $O/ab.err
  timeout 200 python3 bench.py $Q > $O/ab_head_$rep.json 2>> $O/ab.err
done
for v in $(ls gpurun_scratch/liboc_v*.so 2>/dev/null); do
  OC_AMD_LIB=$R/$v timeout 200 python3 bench.py $Q > $O/ab_$(basename $v .so).json 2>> $O/ab.err
done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
for c in 4 5; do
  timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_config$c.json 2>> $O/bench_other.err; echo "config $c rc=$?"
done
timeout 300 python3 bench.py --config 5 --envs 131072 --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_config5_131072.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --envs 131072 --steps 4000 --warmup 400 $Q > $O/bench_cramped_131072.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --envs 1048576 --steps 4000 --warmup 400 $Q > $O/bench_cramped_1M.json 2>> $O/bench_other.err
STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r03 > /dev/null 2>&1
cp gpurun_out/pmc_r03.txt gpurun_out/sq_counters_r03.json $O/ 2>/dev/null
STEPS=4000 timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES -d /tmp/pmc_ic -o p -- python tools/prof_rollout.py > $O/pmc_icache.log 2>&1
STEPS=4000 python tools/pmc_summary.py 4000 $(find /tmp/pmc_ic -name "*.db" -printf "%h\n" | sort -u) -- k_rollout > $O/pmc_icache.txt 2>&1
for f in $O/ab_*.json $O/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print('%.1f G env-steps/s frac %.3f launch_ms %.4f parity %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['launch_ms'], (d.get('parity_check') or {}).get('mismatches')))" 2>&1 | tail -1)"; done
cat $O/pmc_icache.txt | head; cat $O/pmc_r03.txt | tail -3
