#!/bin/bash
# GPU box: time the C2 sweeps with every variant library under single-algebra_amd/lib/exp (+ the default build), twice
# each, and check each variant's sweep against the golden vectors.   bash tools/exp_dq2.sh [names...]
cd $GRAFT_REPO_ROOT
names="$@"; [ -z "$names" ] && names=$(ls single-algebra_amd/lib/exp/ | sed 's/libsapca_//; s/\.so//')
for rep in 1 2; do
  python3 tools/abl_run.py 2>/dev/null | tail -1
  for n in $names; do
    SAPCA_LIB_PATH=$PWD/single-algebra_amd/lib/exp/libsapca_$n.so python3 tools/abl_run.py 2>/dev/null | tail -1
  done
done
for n in $names; do
  echo "== parity $n"
  SAPCA_LIB_PATH=$PWD/single-algebra_amd/lib/exp/libsapca_$n.so timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "g3_spmm_tiled or spmm_tiled_matches or staged_sweep_on_ragged" 2>&1 | tail -2
done
