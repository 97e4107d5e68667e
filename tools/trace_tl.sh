#!/bin/bash
# rocprofv3 --kernel-trace of tools/trace_vecstep.py, then the timeline of the last launches:  tools/trace_tl.sh TAG [n]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl_$1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$1 -- python3 $R/tools/trace_vecstep.py > $R/gpurun_out/tl_$1.log 2>&1 || { tail -5 $R/gpurun_out/tl_$1.log; exit 1; }
f=$(find $R/gpurun_out/tl_$1 -name "*kernel_trace.csv" | tail -1)
python3 $R/tools/trace_timeline.py $f ${2:-40} | tee $R/gpurun_out/tl_$1.txt
python3 $R/tools/trace_gaps.py $f 320 | tail -12
rm -rf $R/gpurun_out/tl_$1
