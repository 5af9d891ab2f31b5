# usage (GPU box): bash tools/timeline.sh  -- kernel timeline (start, duration, stream) of the preparation of the last C2 step
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptl/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last fit: find the last atd_hist / tile_hist start
idx=[i for i,r in enumerate(rows) if "atd_hist" in r["Kernel_Name"] or "pack_rows" in r["Kernel_Name"]]
i0=idx[-1]-2
t0=int(rows[i0]["Start_Timestamp"])
seen=False
for r in rows[i0:i0+70]:
    seen = seen or "dq_info" in r["Kernel_Name"]
    st=(int(r["Start_Timestamp"])-t0)/1e3; du=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("void ","")[:46]
    print("%9.1f us  %8.1f us  q%s  %s" % (st,du,r.get("Queue_Id","?"),name))
    if seen and "spmm_dq" in r["Kernel_Name"]: break
PY
