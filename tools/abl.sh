for mode in 0 1 2 3; do
  SAPCA_TILED_MODE=$mode python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode $mode', 'sweep_ms', round(d['roofline']['avg_launch_ms'],3), 'spmm', round(d['config']['stage_ms']['spmm_ms'],2), 'spmmt', round(d['config']['stage_ms']['spmmt_ms'],2))"
done
