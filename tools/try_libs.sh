for rw in 32 28 24; do
  cp single-algebra_amd/lib/libsapca_rw$rw.so single-algebra_amd/lib/libsapca.so
  for mode in 0 1; do SAPCA_TILED_MODE=$mode python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rw $rw mode $mode', 'ms/step', round(d['ms_per_step'],2), 'sweep_ms', round(d['roofline']['avg_launch_ms'],3), 'spmm', round(d['config']['stage_ms']['spmm_ms']/5,3), 'spmmt', round(d['config']['stage_ms']['spmmt_ms']/5,3))"; done; done
