#!/bin/bash
# rocprofv3 --kernel-trace of one tools/trace_*.py target, then the in-situ duration / gap table:  tools/trace_run.sh TAG SCRIPT [tail]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tr_$1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$1 -- python3 $R/tools/$2 > $R/gpurun_out/tr_$1.log 2>&1 || { tail -5 $R/gpurun_out/tr_$1.log; exit 1; }
f=$(find $R/gpurun_out/tr_$1 -name "*kernel_trace.csv" | tail -1)
python3 $R/tools/trace_gaps.py $f ${3:-300} | tee $R/gpurun_out/tr_$1.txt
rm -rf $R/gpurun_out/tr_$1
