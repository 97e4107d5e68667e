cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01i_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01i_bench_under_rocprof.json 2> $R/gpurun_out/r01i_stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01i_pmc_fetch -- python3 $R/tools/prof_bench_kernels.py > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01i_pmc_write -- python3 $R/tools/prof_bench_kernels.py > /dev/null 2>&1 || exit 1
cd $R
for d in fetch write; do f=$(find gpurun_out/r01i_pmc_$d -name "*counter_collection.csv" | tail -1); echo "== $d"; python tools/pmc_summary.py $f; done
