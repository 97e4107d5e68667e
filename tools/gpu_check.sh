#!/bin/bash
# one GPU call: the whole -m gpu suite, then the in-situ kernel table of the train-only graph and of the full loop
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${1:-chk}_tests.log 2>&1 || { tail -40 gpurun_out/${1:-chk}_tests.log; exit 1; }
tail -3 gpurun_out/${1:-chk}_tests.log
timeout -k 10 200 bash tools/trace_run.sh ${1:-chk}_train trace_trainsteps.py 300 && timeout -k 10 200 bash tools/trace_run.sh ${1:-chk}_loop trace_vecstep.py 300
