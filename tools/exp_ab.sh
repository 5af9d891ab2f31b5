cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('ms/step %.2f  A %.3f  At %.3f  prep %.2f  stats %.2f  transform %.2f' % (d['ms_per_step'], s['spmm_ms']/5, s['spmmt_ms']/5, s['prepare_ms'], s['stats_ms'], s['transform_ms']))"; }
OLD=$GRAFT_REPO_ROOT/single-algebra_amd/lib/exp/libsapca_old.so
NEW=$GRAFT_REPO_ROOT/single-algebra_amd/lib/exp/libsapca_new.so
for i in 1 2; do
run SAPCA_LIB_PATH=$OLD
run SAPCA_LIB_PATH=$NEW

done
