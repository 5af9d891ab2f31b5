#!/bin/bash
# MFMA / VALU counters of the dense kernels (Gram, panel GEMM, Cholesky) over one bench step: tools/pmc_mfma.sh <tag> <workload>
cd /tmp; export TMPDIR=/tmp
tag=$1; wl=${2:-c2}
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> /tmp/pmc_$tag.err
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_$tag/*/*counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
tr=glob.glob("/tmp/pmc_$tag/*/*kernel_trace.csv")
dur={}
for r in csv.DictReader(open(tr[0])): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dsum=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    if not any(s in k for s in ("gram_kernel","gramv_kernel","gram_split_kernel","panel_gemm","chol_inv")): continue
    import re
    k=re.search(r"(gramv_kernel|gram_split_kernel|gram_kernel|panel_gemm_kernel|chol_inv\w*)(<[^>]*>)?", k).group(0)
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    if (k,r["Dispatch_Id"]) not in seen:
        seen.add((k,r["Dispatch_Id"])); dsum[k]+=dur.get(r["Dispatch_Id"],0)
for k in acc:
    n=max(cnt[(k,c)] for c in acc[k])
    print("%s: %d dispatches, avg %.1f us" % (k, n, dsum[k]/n))
    for c,v in acc[k].items(): print("   %-30s %.4g" % (c, v/cnt[(k,c)]))
PY
