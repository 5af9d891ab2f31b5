# usage (GPU box): bash tools/ab_libs.sh <workload> <rounds> <lib name or path> ...   -- interleaved A/B of library builds on ONE box:
# every round runs bench.py once per build; prints every run and the per-build medians.  A name N means lib/exp/libsapca_N.so,
# `default` the tree's build.
cd $GRAFT_REPO_ROOT
WL=$1; ROUNDS=$2; shift 2
rm -f /tmp/ab_libs.txt
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    p=$v
    [ "$v" = default ] && p=$GRAFT_REPO_ROOT/single-algebra_amd/lib/libsapca.so
    [ -f "$p" ] || p=$GRAFT_REPO_ROOT/single-algebra_amd/lib/exp/libsapca_$v.so
    SAPCA_LIB_PATH=$p timeout -k 10 300 python3 bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']
print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], s['prepare_ms'], s['ortho_ms'], s['small_svd_ms'], s['transform_ms'])" >> /tmp/ab_libs.txt
  done
done
python3 - <<PY
import collections, statistics
rows=collections.defaultdict(list)
for ln in open('/tmp/ab_libs.txt'):
    f=ln.split(); rows[f[0]].append([float(x) for x in f[1:]])
print('%-16s %9s %9s %9s %9s %9s %9s   (medians of %d runs; ms)' % ('build','ms/step','sweep','prepare','ortho','small','transform', max(len(v) for v in rows.values())))
for k,v in rows.items():
    print('%-16s' % k, ' '.join('%9.4f' % statistics.median(c) for c in zip(*v)), '   ms/step runs:', ' '.join('%.3f' % r[0] for r in v))
PY
