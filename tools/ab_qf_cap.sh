cd $GRAFT_REPO_ROOT
D=$PWD/single-algebra_amd/lib/libsapca_dbg.so
for wl in c2 c4; do
for r in 1 2 3; do
  for mode in fixed fit; do
    if [ $mode = fixed ]; then export SAPCA_QF_CAP_FIXED=1; else unset SAPCA_QF_CAP_FIXED; fi
    SAPCA_LIB_PATH=$D timeout -k 10 300 python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']
print('$wl $mode', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(s['prepare_ms'],3))"
  done
done
done
