# usage (GPU box): bash tools/timeline_f64.sh  -- kernel timeline of the preparation of one f64 randomized fit on the C2 matrix
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl64 -o tl -- python3 $GRAFT_REPO_ROOT/tools/shape_time.py 200000 20000 0.03 50 10 4 2 f64 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptl64/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "pack_rows64" in r["Kernel_Name"]]
i0=max(0,idx[-1]-8)
t0=int(rows[i0]["Start_Timestamp"])
n=0
for r in rows[i0:]:
    st=(int(r["Start_Timestamp"])-t0)/1e3; du=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("sapca::(anonymous namespace)::","").replace("void ","")[:60]
    if st > 11000: break
    if du>15: print("%9.1f us  %8.1f us  q%s  %s" % (st,du,r.get("Queue_Id","?"),name))
PY
