#!/bin/bash
# per-kernel HIP-event times of the ring-fed B = 32 train step for the product library and every ablation build in build/abl/
R=${GRAFT_REPO_ROOT:-.}
echo "product:"; python3 $R/tools/time_train_ring.py 32 dqn
for f in $R/build/abl/lib_*.so; do echo "$(basename $f):"; FB_LIB=$f python3 $R/tools/time_train_ring.py 32 dqn; done
