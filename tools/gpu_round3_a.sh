#!/bin/bash
# round 3, GPU call A: the gpu suite (incl. the launch-shape tests), the driver's exact bench command (parity_check, same-run
# PMC traffic), the other BASELINE configs with and without the MODE 2 kernels, the reference's own Python timed on this box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r03a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
for c in 4 5; do
  timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 > $O/bench_config$c.json 2>> $O/bench_other.err; echo "config $c rc=$?"
  OC_ROLLOUT_NO_MODE2=1 timeout 300 python3 bench.py --config $c --steps 4000 --warmup 400 --no-parity-check --no-cpu-baseline --no-traffic > $O/bench_config${c}_mode0.json 2>> $O/bench_other.err
done
timeout 300 python3 bench.py --config 5 --envs 131072 --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_config5_131072.json 2>> $O/bench_other.err
OC_ROLLOUT_PIPE=1 timeout 300 python3 bench.py --config 5 --envs 131072 --steps 4000 --warmup 400 --no-cpu-baseline --no-parity-check --no-traffic > $O/bench_config5_131072_pipe1.json 2>> $O/bench_other.err
OC_ROLLOUT_PIPE=0 timeout 300 python3 bench.py --config 5 --steps 4000 --warmup 400 --no-cpu-baseline --no-parity-check --no-traffic > $O/bench_config5_65536_pipe0.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --config 2 --layout asymmetric_advantages --steps 4000 --warmup 400 --no-cpu-baseline --no-extras --no-traffic > $O/bench_asym.json 2>> $O/bench_other.err
timeout 300 python3 bench.py --config 3 --steps 400 --warmup 400 > $O/bench_config3.json 2>> $O/bench_other.err; echo "config 3 rc=$?"
# the reference's own OvercookedEnv.step on this box's host cores (git-ignored tarball made by tools/pack_reference.sh)
if [ -f gpurun_scratch/ref_src.tgz ]; then
  mkdir -p /tmp/ref_src && tar -C /tmp/ref_src -xzf gpurun_scratch/ref_src.tgz
  OVERCOOKED_REFERENCE_SRC=/tmp/ref_src EPISODES=50 timeout 600 python3 tools/time_reference_python.py > $O/reference_python_gpubox.json 2> $O/reference_python.err; echo "reference rc=$?"
fi
# kernel trace of the driver's command
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-traffic > $O/bench_trace.log 2>&1
python3 $R/tools/summarize_prof.py $O $O/r03_driver_cmd_rocprof.txt > /dev/null 2> $O/summarize.err
rm -rf $O/trace
cd $R
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print('%.1f G env-steps/s frac %.3f parity %s' % (d['value']/1e9, d['roofline']['frac'], (d.get('parity_check') or {}).get('mismatches')))" 2>&1 | tail -1)"; done
ls -la $O
