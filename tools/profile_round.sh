# Evidence of one round, on the GPU box:  bash tools/profile_round.sh r03        (then copy gpurun_out/<tag>_* summaries into profiles/)
#   1. rocprofv3 --kernel-trace --stats of the bench command                         -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench kernels      -> <tag>_pmc_fetch_write_raw.json, traffic json
#   3. the per-config table (tools/bench_configs.py)                                  -> <tag>_configs.jsonl
#   4. the plain bench line                                                           -> <tag>_bench.json
#   5. SQ counter passes of the acting and the train kernels (MFMA busy, LDS, waits)  -> <tag>_pmc_sq.txt (MFMA utilisation table on top)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03}
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_pmc_fetch -- python3 $R/tools/prof_bench_kernels.py > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_pmc_write -- python3 $R/tools/prof_bench_kernels.py > /dev/null 2>&1 || exit 1
cd $R
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | tail -1); cp $f gpurun_out/${T}_kernel_stats.csv
python3 tools/update_traffic.py $T > gpurun_out/${T}_traffic.log 2>&1
cp gpurun_out/${T}_traffic.json profiles/traffic.json      # (so that the bench line below carries this build's PMC bytes)
python3 tools/bench_configs.py --steps 100 --cpu --out gpurun_out/${T}_configs.jsonl > gpurun_out/${T}_configs.log 2>&1
python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.json; cat gpurun_out/${T}_configs.jsonl
bash tools/pmc_act.sh ${T}act > gpurun_out/${T}_pmc_sq_act.log 2>&1 && bash tools/pmc_train.sh ${T}train > gpurun_out/${T}_pmc_sq_train.log 2>&1 && {
  echo "# rocprofv3 --pmc (SQ counters): tools/pmc_act.sh (acting forward, 1024 envs) and tools/pmc_train.sh (train step, B = 32, gathered-minibatch form)"
  echo "## MFMA utilisation (tools/mfma_util.py)"
  python3 tools/mfma_util.py $(find gpurun_out/pmc2_${T}act gpurun_out/pmc2_${T}train -name "*counter_collection.csv")
  echo; echo "## per-kernel counter means (tools/pmc_summary.py)"
  grep -hv "^W2026\|^E2026\|amdgpu.ids" gpurun_out/${T}_pmc_sq_act.log gpurun_out/${T}_pmc_sq_train.log
} > gpurun_out/${T}_pmc_sq.txt
