#!/bin/bash
# usage (GPU box): bash tools/final_profiles.sh <round tag, e.g. r02>  -- the bench lines and rocprofv3 summaries kept under profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; out=gpurun_out/final; mkdir -p $out
python3 bench.py > $out/${tag}_c2_bench.json 2> $out/bench.err || exit 1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/${tag}_c2_bench_profiled.json 2>/dev/null ) || exit 1
cp $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats.csv
for wl in c3 c4 c5; do
  timeout -k 10 400 python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/${tag}_${wl}_bench.json 2>/dev/null || exit 1
done
( timeout -k 10 200 python3 tools/masked_random_time.py; timeout -k 10 200 python3 tools/masked_random_time.py f64 ) > $out/${tag}_masked_randomized_c3_matrix.txt 2>/dev/null
timeout -k 10 200 python3 tools/shape_time.py 200000 20000 0.03 50 10 4 3 f64 > $out/${tag}_c2_matrix_f64.txt 2>/dev/null
( echo "== statistics behind the upload (default)"; timeout -k 10 200 python3 tools/host_path_time.py; echo "== SAPCA_UPLOAD_STATS_OFF=1 (a switch of the debug variant: csrc/switches.h)"; SAPCA_LIB_PATH=$GRAFT_REPO_ROOT/single-algebra_amd/lib/libsapca_dbg.so SAPCA_UPLOAD_STATS_OFF=1 timeout -k 10 200 python3 tools/host_path_time.py ) > $out/${tag}_host_path.txt 2>/dev/null
ls -la $out
