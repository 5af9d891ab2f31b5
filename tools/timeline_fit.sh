# usage (GPU box): bash tools/timeline_fit.sh [workload]  -- kernel timeline of the LAST step of a workload: start, duration, idle gap before, queue
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ptlf
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptlf -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${1:-c2} --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptlf/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "atd_hist" in r["Kernel_Name"] or "pack_rows" in r["Kernel_Name"] or "count_kept" in r["Kernel_Name"]]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"])
busy_end=t0; idle=0.0
last_sweep=max(i for i,r in enumerate(rows) if "spmm_dq" in r["Kernel_Name"] or "spmm_row" in r["Kernel_Name"] or "spmv" in r["Kernel_Name"])
for n_,r in enumerate(rows[i0:]):
    if i0+n_ == last_sweep+1:
        print("== one step (first preparation kernel to the end of the projection sweep): %.1f us, GPU idle %.1f us ==" % ((busy_end-t0)/1e3, idle))
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap=max(0,s-busy_end)/1e3; idle+=gap
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("sapca::(anonymous namespace)::","").replace("void ","")[:44]
    print("%9.1f us  %8.1f us  gap %6.1f  q%s  %s" % ((s-t0)/1e3,(e-s)/1e3,gap,r.get("Queue_Id","?"),name))
    busy_end=max(busy_end,e)
print("total %.1f us, idle %.1f us" % ((busy_end-t0)/1e3, idle))
PY
