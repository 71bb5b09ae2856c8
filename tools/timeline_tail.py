#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 --kernel-trace CSV with start/end relative to the first of them (host-tier latency timelines).
usage: timeline_tail.py kernel_trace.csv [n=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bsx::", "").replace("void ", ""), r["Queue_Id"]) for r in rows)[-n:]
t0 = ev[0][0]
for e in ev:
    print("%8.3f -> %8.3f  (%6.3f)  q%s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[2][:48]))
