#!/usr/bin/env python3
"""Steady-state window of a rocprofv3 --kernel-trace CSV: every kernel longer than 30 us between the 8th- and 4th-last big
expansion launches, with start/end relative to the window start.  usage: timeline2.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bsx::", ""), r["Queue_Id"]) for r in rows)
big = [e for e in ev if "expand" in e[2] and e[1] - e[0] > 1_000_000]
t0 = big[-8][0]
for e in ev:
    if t0 - 3_000_000 <= e[0] <= big[-4][1] and e[1] - e[0] > 30_000:
        print("%8.3f -> %8.3f  (%6.3f)  q%s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[2][:40]))
