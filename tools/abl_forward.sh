#!/bin/bash
# HIP-event times of the acting forward's kernels (1024 envs) for the product library and every ablation build in the directory $1
R=${GRAFT_REPO_ROOT:-.}
echo "product:"; python3 $R/tools/time_forward.py 1024 2>/dev/null
for f in $R/${1:-build/abl2}/lib_*.so; do echo "$(basename $f):"; FB_LIB=$f python3 $R/tools/time_forward.py 1024 2>/dev/null; done
