cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
export SPLIT=3 REPS=2
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/$OUT/a -o a -- python3 $R/tools/xgemm_probe.py > /dev/null 2>&1 &&
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES -d $R/gpurun_out/$OUT/b -o b -- python3 $R/tools/xgemm_probe.py > /dev/null 2>&1
ls $R/gpurun_out/$OUT/*
