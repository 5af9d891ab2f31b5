#!/bin/bash
# usage (GPU box): bash tools/pmc_kernel.sh <kernel substring> <workload> "<counters>"  -- averages of rocprofv3 --pmc counters over a kernel's launches in one step
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmck
rocprofv3 --pmc $3 --kernel-trace --output-format csv -d /tmp/pmck -- python3 $GRAFT_REPO_ROOT/bench.py --workload $2 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections,os
f=max(glob.glob("/tmp/pmck/*/*counter_collection.csv"), key=os.path.getmtime)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "$1" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-28s %.4g  x%d" % (k, sum(v)/len(v), len(v)))
PY
