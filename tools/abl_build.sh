#!/bin/bash
# Builds experiment variants of libsapca.so that differ only in spmm_tiled.hip's compile-time switches:
#   tools/abl_build.sh NAME "-DSAPCA_ABL=1 ..."   ->  single-algebra_amd/lib/exp/libsapca_NAME.so
# Run one with SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../single-algebra_amd"
name=$1; shift
mkdir -p build/exp lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $@ -c csrc/spmm_tiled.hip -o build/exp/spmm_tiled_$name.o
objs=$(ls build/*.o | grep -v spmm_tiled.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/exp/libsapca_$name.so $objs build/exp/spmm_tiled_$name.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built lib/exp/libsapca_$name.so
