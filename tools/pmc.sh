#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>"   -> gpurun_out/pmc_<tag>/ ; prints per-kernel averages for sapca kernels
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$tag.err
python - <<EOF
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv")
if not f: print("no counter file", glob.glob("gpurun_out/pmc_$tag/*/*")); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "sapca" not in k: continue
    k=k.split("(")[0][-60:]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    cnt[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k)
    for c,v in acc[k].items(): print("   ",c, v/cnt[(k,c)], "x",cnt[(k,c)])
EOF
