# usage (GPU box): bash tools/abl.sh "abl0 abl1 ..." "0 8 10"   -- variants built by tools/abl_build.sh x SAPCA_TILED_MODE values
for v in $1; do for mode in $2; do
  SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_$v.so SAPCA_TILED_MODE=$mode timeout -k 10 120 python tools/abl_run.py 2>/dev/null || exit 1
done; done
