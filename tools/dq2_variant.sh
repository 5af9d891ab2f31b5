#!/bin/bash
# Builds an experiment variant of libsapca.so that differs in the generated main loop of the DPP-fed sweep (and, with
# CXXDEFS, in compile-time switches of spmm_dq.hip / spmm_tiled.hip):
#   tools/dq2_variant.sh NAME [ENV=VALUE ...]   ->  single-algebra_amd/lib/exp/libsapca_NAME.so
# The switches are those of tools/gen_spmm_dq2.py (DQ2_B64=1, DQ2_FMAC=1, DQ2_DEPTH=3, DQ2_PRIO=1, DQ2_ODD=1);
# CXXDEFS="-DSAPCA_EVEN_STEPS" goes to the compiler (DQ2_ODD=0 needs it: format and main loop must agree).  Run one with
# SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_NAME.so python3 tools/abl_run.py
set -e
cd "$(dirname "$0")/../single-algebra_amd"
name=$1; shift
mkdir -p build/exp lib/exp
defs=""
for kv in "$@"; do case $kv in CXXDEFS=*) defs="${kv#CXXDEFS=}";; esac; done
env "$@" python3 ../tools/gen_spmm_dq2.py build/exp/dq2_gen_$name.h > /dev/null
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $defs"
/opt/rocm/bin/hipcc $flags -DDQ2_GEN_H="\"$PWD/build/exp/dq2_gen_$name.h\"" -c csrc/spmm_dq.hip -o build/exp/spmm_dq_$name.o
skip="build/spmm_dq.o"
extra="build/exp/spmm_dq_$name.o"
if [ -n "$defs" ]; then
  /opt/rocm/bin/hipcc $flags -c csrc/spmm_tiled.hip -o build/exp/spmm_tiled_$name.o
  skip="$skip|build/spmm_tiled.o"
  extra="$extra build/exp/spmm_tiled_$name.o"
fi
objs=$(ls build/*.o | grep -v -E "$skip")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/exp/libsapca_$name.so $objs $extra -ldl -Wl,-rpath,/opt/rocm/lib
echo built lib/exp/libsapca_$name.so
