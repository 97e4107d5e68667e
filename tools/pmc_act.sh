# SQ counters of the acting-forward kernels (1024 envs): bash tools/pmc_act.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-act}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_$T -- python3 $R/tools/act_once.py > $R/gpurun_out/pmc_$T.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2_$T -- python3 $R/tools/act_once.py >> $R/gpurun_out/pmc_$T.txt 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $R/gpurun_out/pmc3_$T -- python3 $R/tools/act_once.py >> $R/gpurun_out/pmc_$T.txt 2>&1 || echo "pass 3 failed"
cd $R
for d in pmc_$T pmc2_$T pmc3_$T; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | tail -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "conv1_sp|conv23_sp|fc1_sp|env_kernel"; done
