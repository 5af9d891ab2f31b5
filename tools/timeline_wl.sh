# usage (GPU box): bash tools/timeline_wl.sh c4|c5  -- kernel timeline (start, duration, stream) of the preparation of the last step of a workload
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptlw -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptlw/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last step: from 60 kernels before its last tile_hist to its first sweep
idx=[i for i,r in enumerate(rows) if "tile_hist" in r["Kernel_Name"]]
i0=max(0,idx[-1]-60)
t0=int(rows[i0]["Start_Timestamp"])
seen=False
for i,r in enumerate(rows[i0:i0+200]):
    st=(int(r["Start_Timestamp"])-t0)/1e3; du=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("void ","")[:46]
    if du > 30: print("%9.1f us  %8.1f us  q%s  %s" % (st,du,r.get("Queue_Id","?"),name))
    if i0+i > idx[-1] and "spmm_dq" in r["Kernel_Name"]: break
PY
