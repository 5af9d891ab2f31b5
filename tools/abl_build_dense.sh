#!/bin/bash
# like abl_build.sh, for csrc/dense.hip:  tools/abl_build_dense.sh NAME "-DFLAG=1"
set -e
cd "$(dirname "$0")/../single-algebra_amd"
name=$1; shift
mkdir -p build/exp lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $@ -c csrc/dense.hip -o build/exp/dense_$name.o
objs=$(ls build/*.o | grep -v "build/dense.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/exp/libsapca_$name.so $objs build/exp/dense_$name.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built lib/exp/libsapca_$name.so
