#!/usr/bin/env python3
"""Steady-state window of a rocprofv3 --kernel-trace CSV: every kernel that starts within [a, b] ms before the last kernel's
end, with start/end relative to the window start.  usage: timeline3.py kernel_trace.csv [a_ms=8] [b_ms=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
a = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
b = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bsx::", "").replace("void ", ""), r["Queue_Id"]) for r in rows)
ours = [e for e in ev if e[2].startswith("k_")]
tend = ours[-1][1]
t0 = tend - int(a * 1e6)
for e in ours:
    if t0 <= e[0] <= tend - int(b * 1e6):
        print("%8.3f -> %8.3f  (%6.3f)  q%s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[2][:40]))
