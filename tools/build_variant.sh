#!/bin/bash
# build_ab/lib_<name>.so = the in-tree objects with ONE source recompiled under extra flags (diagnostic builds: A/B on one box
# through RAGRAPH_HIP_SO).   tools/build_variant.sh smalltiming topk_small.hip -DRG_SMALL_TIMING
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build_ab
cd $R/ragraph_amd/csrc
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wall -Wno-unused-function -fno-fast-math -ffp-contract=on "$@" -c $src -o $R/build_ab/${src%.hip}_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}\.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build_ab/${src%.hip}_$name.o -o $R/build_ab/lib_$name.so
echo built $R/build_ab/lib_$name.so
