cd $GRAFT_REPO_ROOT
for wl in c4 c5; do
for sw in 0 1; do
  if [ $sw = 1 ]; then export SAPCA_AT_SORT=1; else unset SAPCA_AT_SORT; fi
  timeout -k 10 400 python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('$wl sort=$sw ms/step %.2f  prep %.2f stats %.2f  A %.2f At %.2f ortho %.2f transform %.2f' % (d['ms_per_step'], s['prepare_ms'], s['stats_ms'], s['spmm_ms'], s['spmmt_ms'], s['ortho_ms'], s['transform_ms']))"
done; done
