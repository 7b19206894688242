#!/bin/bash
# round 3, GPU call C: gpu suite on the restructured MODE 1 step (look-ups / shadow / interacts), A/B against variants.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r03c}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
Q="--no-extras --no-cpu-baseline --no-traffic"
for rep in 1 2; do
  [ -f gpurun_scratch/liboc_r2.so ] && OC_AMD_LIB=$R/gpurun_scratch/liboc_r2.so timeout 200 python3 bench.py $Q --no-parity-check > $O/ab_r2_$rep.json 2>> $O/ab.err
  timeout 200 python3 bench.py $Q > $O/ab_head_$rep.json 2>> $O/ab.err
  for v in $(ls gpurun_scratch/liboc_v*.so 2>/dev/null); do
    OC_AMD_LIB=$R/$v timeout 200 python3 bench.py $Q > $O/ab_$(basename $v .so)_$rep.json 2>> $O/ab.err
  done
done
for c in 4 5; do
  timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 --no-cpu-baseline --no-traffic > $O/bench_config$c.json 2>> $O/bench_other.err; echo "config $c rc=$?"
done
timeout 300 python3 bench.py --envs 131072 --steps 4000 --warmup 400 $Q > $O/bench_cramped_131072.json 2>> $O/bench_other.err
STEPS=4000 timeout 600 bash tools/pmc_rollout.sh r03c > /dev/null 2>&1
cp gpurun_out/pmc_r03c.txt gpurun_out/sq_counters_r03c.json $O/ 2>/dev/null
for f in $O/ab_*.json $O/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print('%.1f G env-steps/s frac %.3f launch_ms %.4f parity %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['launch_ms'], (d.get('parity_check') or {}).get('mismatches')))" 2>&1 | tail -1)"; done
tail -3 $O/pmc_r03c.txt
