#!/bin/bash
# ablation builds of ONE source file:  tools/abl_build.sh OUTDIR FILE(.hip stem) "-DKNOB=1" "-DKNOB=2" ...   (runs here: hipcc cross-compiles)
# -> OUTDIR/lib_<flag>.so = the product's other objects + FILE rebuilt with the flag
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/dqnflappybird_amd/csrc; O=$R/$1; F=$2; shift 2
mkdir -p $O
others=$(ls $C/*.o | grep -v "/$F.o")
for flag in "$@"; do
    tag=$(echo "$flag" | tr -d ' -' | tr '=' '_' | sed 's/^D//')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $flag -c $C/$F.hip -o $O/${F}_$tag.o &
done
wait
for flag in "$@"; do
    tag=$(echo "$flag" | tr -d ' -' | tr '=' '_' | sed 's/^D//')
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $O/lib_$tag.so $others $O/${F}_$tag.o
    echo $O/lib_$tag.so
done
