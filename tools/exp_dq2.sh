cd /tmp; export TMPDIR=/tmp
for mode in 0 2 4 6; do
  rm -rf /tmp/pr_$mode
  SAPCA_DQ_MODE=$mode timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== mode $mode"
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pr_$mode/*/*kernel_trace.csv")
import collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "spmm_dq" in r["Kernel_Name"]:
        d[r["Grid_Size_X"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for g,v in d.items(): print("  grid", g, "n", len(v), "avg ms %.3f" % (sum(v)/len(v)), "min %.3f" % min(v))
PY
done
