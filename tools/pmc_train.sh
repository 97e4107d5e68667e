# SQ counters of the train-step kernels (B = 32): bash tools/pmc_train.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-sq}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_$T -- python3 $R/tools/time_train.py 32 dqn > $R/gpurun_out/pmc_$T.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2_$T -- python3 $R/tools/time_train.py 32 dqn >> $R/gpurun_out/pmc_$T.txt 2>&1 || exit 1
cd $R
for d in pmc_$T pmc2_$T; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | tail -1); python tools/pmc_summary.py $f; done
