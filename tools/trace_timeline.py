#!/usr/bin/python3
"""Timeline of ONE cycle out of a rocprofv3 --kernel-trace CSV: every dispatch between two consecutive launches of the coarse
tail kernel (one per V-cycle), with its start offset, duration, queue and the idle time in front of it on the whole GPU.
    python tools/trace_timeline.py <rocprof_out_dir> [cycle_index=20] [marker=cycle3d_tail_kernel]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
marker = sys.argv[3] if len(sys.argv) > 3 else "cycle3d_tail_kernel"
rows = []
for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[k], marks[k + 1]
t0 = int(rows[a]["End_Timestamp"])
print("cycle %d: %d dispatches, %.3f ms from the end of one tail kernel to the end of the next" % (k, b - a, (int(rows[b]["End_Timestamp"]) - t0) / 1e6))
prev_end = t0
busy = {}
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void mgx::", "").split("(")[0][:58]
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    q = r.get("Queue_Id", "?")
    gap = s - prev_end
    print("%9.1f us  +%7.1f us  q%-3s %-58s grid %9d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name, grid, ("idle %.1f" % (gap / 1e3)) if gap > 1500 else ""))
    prev_end = max(prev_end, e)
    busy[q] = busy.get(q, 0) + e - s
print("busy per queue (us):", {q: round(v / 1e3, 1) for q, v in busy.items()})
