cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/patd -o atd -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/atd.json 2>/dev/null
python3 - <<PY
import csv,glob,json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/atd.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["stage_ms"])
f=glob.glob("/tmp/patd/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("atd_","quad_fill","quad_count","tile_hist","dq_desc","trampoline","row_lengths","natural_quad")):
        print(n[:80], r["Calls"], "%.3f ms/fit" % (float(r["TotalDurationNs"])/1e6/4), "avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
