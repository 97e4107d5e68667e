#!/bin/bash
# rocprofv3 --kernel-trace of tools/dbg_per_slow.py MODE, then the timeline of the last launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl_dbg_$1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_dbg_$1 -- python3 $R/tools/dbg_per_slow.py $1 > $R/gpurun_out/tl_dbg_$1.log 2>&1 || { tail -5 $R/gpurun_out/tl_dbg_$1.log; exit 1; }
f=$(find $R/gpurun_out/tl_dbg_$1 -name "*kernel_trace.csv" | tail -1)
python3 $R/tools/trace_timeline.py $f ${2:-40} > $R/gpurun_out/tl_dbg_$1.txt
tail -1 $R/gpurun_out/tl_dbg_$1.log
rm -rf $R/gpurun_out/tl_dbg_$1
