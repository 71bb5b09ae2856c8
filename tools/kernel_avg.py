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
    if not big:
        continue
    # a kernel launched at two very different sizes (k_expand_witness: 4096 map jobs vs the handful of reduce nodes) is
    # reported per size class, so that the large launches' average can be compared with bench.py's roofline.avg_launch_ms
    top = [x for x in big if x > 0.5 * max(big)]
    rest = [x for x in big if x <= 0.5 * max(big)]
    if rest and len(top) >= 2 and max(big) > 5 * min(big):
        print("%-40s n=%4d  avg %.3f ms  max %.3f   (large launches)" % (n[:40], len(top), sum(top) / len(top), max(top)))
        print("%-40s n=%4d  avg %.3f ms  max %.3f   (small launches)" % (n[:40], len(rest), sum(rest) / len(rest), max(rest)))
    else:
        print("%-40s n=%4d  avg %.3f ms  max %.3f" % (n[:40], len(big), sum(big) / len(big), max(big)))
