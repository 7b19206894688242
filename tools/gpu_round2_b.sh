#!/bin/bash
# round 2, GPU call B: full gpu suite, the driver's exact bench command + rocprof (trace, FETCH/WRITE) of it, SQ counters,
# 1-rank RCCL smoke, the other BASELINE configs.  Everything lands in gpurun_out/r02b/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${OUT_TAG:-r02b}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -2 $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
python3 bench.py --no-extras --no-cpu-baseline > $O/bench_default.json 2>> $O/bench_driver_cmd.err
OC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_force_dist_nccl.json 2> $O/bench_force_dist_nccl.err; echo "nccl rc=$?"
for c in 3 4 5; do python3 bench.py --config $c --steps 400 --warmup 400 > $O/bench_config$c.json 2>> $O/bench_driver_cmd.err; done
python3 bench.py --envs 1048576 --no-extras --no-cpu-baseline --steps 400 --warmup 400 > $O/bench_1M_envs.json 2>> $O/bench_driver_cmd.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_write.log 2>&1
python $R/tools/summarize_prof.py $O $O/r02_driver_cmd_rocprof.txt > /dev/null 2>$O/summarize.err
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
# configs[2] through oc_rollout_encode: kernel trace of the same command (WRITE_SIZE wraps on its multi-GB launches: not collected)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3trace -o c3 -- python3 $R/bench.py --config 3 --steps 400 --warmup 400 > $O/bench_config3_trace.log 2>&1
find /tmp/c3trace -name "*kernel_stats.csv" -exec cp {} $O/r02_config3_kernel_stats.csv \;
cd $R
STEPS=4000 bash tools/pmc_rollout.sh r02 > /dev/null 2>&1
cp gpurun_out/pmc_r02.txt gpurun_out/sq_counters_r02.json $O/ 2>/dev/null
ls -la $O
head -c 600 $O/bench_driver_cmd.json
timeout 240 python tools/soak.py --seeds 12 --envs 4096 --steps 500 > $O/soak.log 2>&1; echo "soak rc=$?" >> $O/soak.log; tail -3 $O/soak.log
