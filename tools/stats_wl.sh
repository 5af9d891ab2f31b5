# usage (GPU box): bash tools/stats_wl.sh c4|c5|c3  -- rocprofv3 kernel statistics of two steps of a workload (top 25 by total time)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pstw -o st -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pstw/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "sapca" in r["Name"] or "rocprim" in r["Name"]]
for r in rows[:25]:
    print(r["Name"].replace("sapca::k::(anonymous namespace)::","").replace("sapca::(anonymous namespace)::","").replace("void ","")[:58].ljust(58), r["Calls"].rjust(5), ("%.1f" % (float(r["TotalDurationNs"])/1e6)).rjust(9), "ms  avg", ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(9), "us")
PY
