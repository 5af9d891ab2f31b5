#!/bin/bash
# HBM traffic of the format builders: rocprofv3 --pmc passes over bench.py --steps 1 (FETCH_SIZE and WRITE_SIZE separately)
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcp_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcp_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmcp_$c/*/*counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    for k in ("atd_scatter","atd_fill","atd_hist","quad_fill_staged","tile_hist","quad_count","dq_desc","atd_stats"):
        if k in n: acc[k].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("$c %-18s %d launches, avg %.1f MB (x2 for FETCH on gfx950: %.1f MB)" % (k,len(v),sum(v)/len(v)/1024, 2*sum(v)/len(v)/1024))
PY
done
