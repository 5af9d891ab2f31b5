#!/usr/bin/env python3
"""GPU idle gaps inside one bench step, from a rocprofv3 --kernel-trace CSV: python3 tools/gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
# last step = from the last transpose/pack kernel onwards
starts = [i for i, e in enumerate(ev) if "pack_rows_permuted" in e[2] or "radix" in e[2].lower() and False]
i0 = starts[-1] if starts else 0
ev = ev[i0:]
t0 = ev[0][0]
busy_end = ev[0][1]
tot_gap = 0.0
def short(n):
    import re
    m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return (m.group(1) if m else n)[:40]
for k in range(1, len(ev)):
    s, e, n = ev[k]
    gap = (s - busy_end) / 1e3
    if gap > thr:
        print("t=%8.1f us  gap %7.1f us  before %-40s after %s" % ((s - t0) / 1e3, gap, short(n), short(ev[k - 1][2])))
    if gap > 0:
        tot_gap += gap
    busy_end = max(busy_end, e)
print("step span %.1f us, idle %.1f us, %d kernels" % ((busy_end - t0) / 1e3, tot_gap, len(ev)))
