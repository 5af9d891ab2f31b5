#!/bin/bash
# HBM traffic of the sweep kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share one) over bench.py --steps 1
# usage (GPU box): bash tools/pmc_traffic.sh <tag> [workload]  -> gpurun_out/<tag>_pmc_{fetch,write}_size.csv + a summary on stdout
cd /tmp; export TMPDIR=/tmp
tag=$1; wl=${2:-c2}
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rm -rf /tmp/pmc_${tag}_$lc
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$lc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections,re
import os
f=max(glob.glob("/tmp/pmc_${tag}_$lc/*/*counter_collection.csv"), key=os.path.getmtime)
rows=[r for r in csv.DictReader(open(f)) if "spmm_dq" in r["Kernel_Name"] or "spmm_quad" in r["Kernel_Name"] or "spmm_rowgather" in r["Kernel_Name"]]
out=open("$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$lc.csv","w")
w=csv.writer(out); w.writerow(["Dispatch_Id","Kernel","Grid_Size","Counter_Name","Counter_Value"])
acc=collections.defaultdict(list)
for r in rows:
    k=re.search(r"(spmm_\w+)(<[^>]*>)?", r["Kernel_Name"]).group(0)
    w.writerow([r["Dispatch_Id"],k,r["Grid_Size"],r["Counter_Name"],r["Counter_Value"]])
    acc[(k,r["Grid_Size"])].append(float(r["Counter_Value"]))
out.close()
for (k,g),v in acc.items(): print("$c %s grid %s: %d launches, avg %.0f KB" % (k,g,len(v),sum(v)/len(v)))
PY
done
