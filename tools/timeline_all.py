#!/usr/bin/env python3
"""Kernels AND memory copies of the last `a` ms of a rocprofv3 run as one text timeline: timeline_all.py kernel_trace.csv memory_copy_trace.csv [a_ms=2]"""
import csv, sys
a = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("bsx::", "").replace("void ", "")[:44])))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
tend = ev[-1][1]
t0 = tend - int(a * 1e6)
for s, e, n in ev:
    if s >= t0:
        print("%8.3f -> %8.3f  (%6.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
