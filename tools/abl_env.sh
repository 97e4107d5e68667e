#!/bin/bash
# in-situ duration of env_kernel<true> (fb_vec_step, 1024 envs) and the env-only rates for the product and every build in $1 (tools/abl_build.sh ... fb_env -D...)
R=${GRAFT_REPO_ROOT:-.}
echo "product:"; bash $R/tools/trace_run.sh envp trace_vecstep.py 400 | grep -E "env_kernel|launches over"; python3 $R/tools/bench_env.py 2>/dev/null | grep "u8=False"
for f in $R/${1:-build/abl_env}/lib_*.so; do echo "$(basename $f):"; FB_LIB=$f bash $R/tools/trace_run.sh env_$(basename $f .so) trace_vecstep.py 400 | grep -E "env_kernel|launches over"; FB_LIB=$f python3 $R/tools/bench_env.py 2>/dev/null | grep "u8=False"; done
