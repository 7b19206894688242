cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export OC_ENC_LDS=${OC_ENC_LDS:-20480}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $R/gpurun_out/pmc_e1 -o e1 -- python gpurun_scratch/enc_only.py > /tmp/l1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_e2 -o e2 -- python gpurun_scratch/enc_only.py > /tmp/l2.log 2>&1
tail -2 /tmp/l2.log | cut -c1-200; du -sh $R/gpurun_out
