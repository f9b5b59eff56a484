#!/usr/bin/python3
"""Per-kernel / per-grid-size breakdown of a rocprofv3 --kernel-trace CSV, plus GPU idle time between kernels.
    python tools/trace_breakdown.py <rocprof_out_dir> [--skip N]   (skip the first N dispatches: set-up, warm-up)"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = 0
for a in sys.argv[2:]:
    if a.startswith("--skip="):
        skip = int(a[7:])
rows = []
for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
# the pipelined smoother is launched with one workgroup per CU on every large level, so dispatches of different
# levels share a grid size: split such a group where its durations are clearly bimodal
byk = collections.defaultdict(list)
for r in rows:
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    byk[(r["Kernel_Name"], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cut = {k: (min(v) * max(v)) ** 0.5 for k, v in byk.items() if max(v) > 3 * min(v)}
agg = collections.OrderedDict()
busy = 0
gaps = 0
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    name = r["Kernel_Name"].replace("void mgx::", "").split("(")[0][:60]
    c = cut.get((r["Kernel_Name"], grid))
    if c:
        name = name[:52] + (" [long]" if e - s >= c else " [short]")
    k = (name, grid)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
    busy += e - s
    if prev_end is not None and s > prev_end:
        gaps += s - prev_end
    prev_end = max(prev_end or e, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("dispatches %d  span %.3f ms  kernel-busy %.3f ms  idle gaps %.3f ms" % (len(rows), span / 1e6, busy / 1e6, gaps / 1e6))
for (name, grid), (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s grid %10d  calls %5d  total %9.3f ms  avg %8.2f us  %5.1f%%" % (name, grid, calls, ns / 1e6, ns / calls / 1e3, 100.0 * ns / span))
