#!/bin/bash
# round 3, GPU call D: gpu suite (regen tests, restored multi-agent test, drop-in surface), headline A/B, lean regime.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r03d}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
Q="--no-extras --no-cpu-baseline --no-traffic"
[ -f gpurun_scratch/liboc_r2.so ] && OC_AMD_LIB=$R/gpurun_scratch/liboc_r2.so timeout 200 python3 bench.py $Q --no-parity-check > $O/ab_r2_1.json 2>> $O/ab.err
timeout 200 python3 bench.py $Q > $O/ab_head_1.json 2>> $O/ab.err
for v in $(ls gpurun_scratch/liboc_v*.so 2>/dev/null); do
  OC_AMD_LIB=$R/$v timeout 200 python3 bench.py $Q > $O/ab_$(basename $v .so)_1.json 2>> $O/ab.err
done
timeout 300 python3 bench.py --envs 131072 --steps 4000 --warmup 400 $Q > $O/bench_cramped_131072.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --envs 1048576 --steps 4000 --warmup 400 $Q --no-parity-check > $O/bench_cramped_1M.json 2>> $O/bench_other.err
for c in 4 5; do
  timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 --no-cpu-baseline --no-traffic > $O/bench_config$c.json 2>> $O/bench_other.err
done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
for f in $O/ab_*.json $O/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print('%.1f G env-steps/s frac %.3f launch_ms %.4f parity %s' % (d['value']/1e9, d['roofline']['frac'], d['roofline']['launch_ms'], (d.get('parity_check') or {}).get('mismatches')))" 2>&1 | tail -1)"; done
python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json'))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in('value','us_per_step','launch_ms','obs_u8','obs_f32')}) for k,v in d.items() if k in ('step_api','single_env_api','training_env')})"
