#!/bin/bash
# the rocprofv3 --kernel-trace --stats pass of the bench command alone (tools/profile_round.sh step 1) + the plain bench line:  tools/profile_stats.sh r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r04}; O=$R/gpurun_out
rm -rf $O/${T}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $O/${T}_bench_under_rocprof.json 2> $O/${T}_stats.err || exit 1
cd $R
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | tail -1); cp $f gpurun_out/${T}_kernel_stats.csv
python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.json
