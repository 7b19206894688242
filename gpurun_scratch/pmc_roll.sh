cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in lane_per_env table_interact; do
MODE=$m rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $R/gpurun_out/pmc_$m -o p -- python gpurun_scratch/roll_only.py > /tmp/l1.log 2>&1
MODE=$m rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $R/gpurun_out/pmc2_$m -o p -- python gpurun_scratch/roll_only.py > /tmp/l1.log 2>&1
done
