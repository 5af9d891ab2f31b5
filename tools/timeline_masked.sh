# usage (GPU box): bash tools/timeline_masked.sh -- kernel timeline of one masked randomized preparation (C3's matrix in f32)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptlm -o tl -- python3 $GRAFT_REPO_ROOT/tools/masked_random_time.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ptlm/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "count_kept" in r["Kernel_Name"]]
i0=idx[-2] if len(idx)>1 else idx[-1]
# the last fit's first compaction: two count_kept per fit (kept, dropped)
t0=int(rows[i0]["Start_Timestamp"])
seen=False
for r in rows[i0:i0+120]:
    st=(int(r["Start_Timestamp"])-t0)/1e3; du=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    name=r["Kernel_Name"].replace("sapca::k::(anonymous namespace)::","").replace("void ","")[:50]
    if du>15: print("%9.1f us  %8.1f us  q%s  %s" % (st,du,r.get("Queue_Id","?"),name))
    if "spmm_dq" in r["Kernel_Name"]: break
PY
