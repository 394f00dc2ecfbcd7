#!/usr/bin/env python3
"""Prints the kernels of two steady-state steps from a rocprofv3 kernel trace (start, end, hardware
queue, name, duration), to see what overlaps with what.  Usage: tools/trace_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("orbx::", "").replace("void ", "")
    if n.startswith("k_") and not n.startswith(("k_knn", "k_top2")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Queue_Id"])))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2] == "k_pyr_level0"]
i0, i1 = idx[len(idx) // 2], idx[min(len(idx) // 2 + 4, len(idx) - 1)]
base = ev[i0][0]
busy = 0
last = base
for e in ev[i0:i1]:
    print("%8.1f %8.1f  q%d %-18s %7.1f us" % ((e[0] - base) / 1e3, (e[1] - base) / 1e3, e[3], e[2][:18], (e[1] - e[0]) / 1e3))
print("span %.1f us" % ((ev[i1][0] - base) / 1e3))
