# usage (GPU box): bash tools/ab_r4.sh [workload]   -- A/B on ONE box: lib/exp/libsapca_r3.so (round 3's HEAD) against the tree's build
cd $GRAFT_REPO_ROOT
WL=${1:-c2}
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('ms/step %.3f  sweep %.4f  prep %.2f  ortho %.3f  small %.3f  transform %.3f  stats %.2f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], s['prepare_ms'], s['ortho_ms'], s['small_svd_ms'], s['transform_ms'], s['stats_ms']))"; }
OLD=$GRAFT_REPO_ROOT/single-algebra_amd/lib/exp/libsapca_r3.so
for i in 1 2 3; do
run SAPCA_LIB_PATH=$OLD
run SAPCA_X=1
done
