# usage (GPU box): bash tools/timeline_c3.sh -- kernel timeline of the preparation and projection of one C3 step (masked Lanczos, f64)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl3 -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptl3/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "count_kept" in r["Kernel_Name"]]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"])
n=0
for r in rows[i0:]:
    st=(int(r["Start_Timestamp"])-t0)/1e3; du=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("sapca::(anonymous namespace)::","").replace("void ","")[:60]
    if "spmv" in name or "gemv" in name or "norm2" in name or "scale" in name or "combine" in name:
        n+=1
        if n>6: continue
    if du>20 or n<=6: print("%9.1f us  %8.1f us  q%s  %s" % (st,du,r.get("Queue_Id","?"),name))
PY
