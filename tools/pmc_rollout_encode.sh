#!/bin/bash
# SQ counter passes over tools/time_rollout_encode.py (run on the GPU box); summary -> gpurun_out/pmc_rollout_encode.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R && export TMPDIR=/tmp
O=/tmp/pmc_re
LAY=${1:-asymmetric_advantages}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/a -o p -- python tools/time_rollout_encode.py $LAY 65536 20 > /tmp/l1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $O/b -o p -- python tools/time_rollout_encode.py $LAY 65536 20 > /tmp/l2.log 2>&1
mkdir -p $R/gpurun_out
python tools/pmc_summary.py $(find $O -name "*.db" -printf "%h\n" | sort -u) -- k_rollout_encode > $R/gpurun_out/pmc_rollout_encode_$LAY.txt 2>&1
cat $R/gpurun_out/pmc_rollout_encode_$LAY.txt
