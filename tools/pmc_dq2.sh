#!/bin/bash
# usage (GPU box): bash tools/pmc_dq2.sh <tag> "<counters>" -- one rocprofv3 --pmc pass over bench.py --steps 1; per-kernel averages of the
# sweep kernels' counters and durations (the effective clock of a dispatch is GRBM_GUI_ACTIVE / 8 / duration)
cd /tmp; export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmc_$tag.err
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_$tag/*/*counter_collection.csv")
kt=glob.glob("/tmp/pmc_$tag/*/*kernel_trace.csv")
if not f: print("no counter file"); raise SystemExit
dur={}
if kt:
    for r in csv.DictReader(open(kt[0])): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if "spmm_dq" not in k and "spmm_quad" not in k: continue
    k=k.split("(")[0][-40:]+" grid "+r.get("Grid_Size","?")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    if (k,r["Dispatch_Id"]) not in seen and r["Dispatch_Id"] in dur:
        seen.add((k,r["Dispatch_Id"])); acc[k]["duration_us"]+=dur[r["Dispatch_Id"]]; cnt[(k,"duration_us")]+=1
for k in acc:
    print(k)
    for c,v in acc[k].items(): print("   %-28s %.6g  x%d" % (c, v/cnt[(k,c)], cnt[(k,c)]))
PY
