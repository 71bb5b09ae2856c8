#!/usr/bin/env python3
"""Average / count of every bsx kernel's duration in a rocprofv3 --kernel-trace CSV (steady state: second half of the run)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
acc = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("bsx::", "").replace("void ", "")
    acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x > 0.02]
    if big:
        print("%-40s n=%4d  avg %.3f ms  max %.3f" % (n[:40], len(big), sum(big) / len(big), max(big)))
