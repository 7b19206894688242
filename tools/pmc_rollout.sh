#!/bin/bash
# SQ counter passes over tools/prof_rollout.py (run on the GPU box): tools/pmc_rollout.sh <tag>; summary -> gpurun_out/pmc_<tag>.txt
TAG=${1:-roll}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R && export TMPDIR=/tmp
O=/tmp/pmc_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/a -o p -- python tools/prof_rollout.py > /tmp/l1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $O/b -o p -- python tools/prof_rollout.py > /tmp/l2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT -d $O/c -o p -- python tools/prof_rollout.py > /tmp/l3.log 2>&1
mkdir -p $R/gpurun_out
python tools/pmc_summary.py --json $R/gpurun_out/sq_counters_$TAG.json ${STEPS:-100} $(find $O -name "*.db" -printf "%h\n" | sort -u) -- k_rollout > $R/gpurun_out/pmc_$TAG.txt 2>&1
cat $R/gpurun_out/pmc_$TAG.txt
