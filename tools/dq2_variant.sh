#!/bin/bash
# Builds an experiment variant of libsapca.so that differs only in the generated main loop of the DPP-fed sweep:
#   tools/dq2_variant.sh NAME [ENV=VALUE ...]   ->  single-algebra_amd/lib/exp/libsapca_NAME.so
# (the switches are those of tools/gen_spmm_dq2.py, e.g. DQ2_B64=1 DQ2_FMAC=1).  Run one with
# SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_NAME.so python3 tools/abl_run.py
set -e
cd "$(dirname "$0")/../single-algebra_amd"
name=$1; shift
mkdir -p build/exp lib/exp
env "$@" python3 ../tools/gen_spmm_dq2.py build/exp/dq2_gen_$name.h > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DDQ2_GEN_H="\"$PWD/build/exp/dq2_gen_$name.h\"" -c csrc/spmm_dq.hip -o build/exp/spmm_dq_$name.o
objs=$(ls build/*.o | grep -v "build/spmm_dq.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/exp/libsapca_$name.so $objs build/exp/spmm_dq_$name.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built lib/exp/libsapca_$name.so
