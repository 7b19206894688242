#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic of bench.py.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $BENCH > $OUT/bench_write.log 2>&1
# summarise on the box and keep only the text / json (the rocpd databases exceed what gpurun_out carries back)
python $ROOT/tools/summarize_prof.py $OUT $OUT/${TAG}_bench_rocprof.txt > /dev/null
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
