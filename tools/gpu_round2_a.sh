#!/bin/bash
# round 2, GPU call A: gpu tests, the driver's exact bench command (+ rocprof of it), nccl single-rank smoke
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
python3 bench.py --no-extras --no-cpu-baseline > $O/bench_default.json 2>> $O/bench_driver_cmd.err
OC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_force_dist_nccl.json 2> $O/bench_force_dist_nccl.err; echo "nccl rc=$?"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_write.log 2>&1
python $R/tools/summarize_prof.py $O $O/r02a_bench_rocprof.txt > /dev/null 2>$O/summarize.err
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
ls -la $O
head -c 1500 $O/bench_driver_cmd.json
