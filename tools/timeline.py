#!/usr/bin/env python3
"""Print one step's kernel timeline (start offset, duration, name) from a rocprofv3 --kernel-trace CSV.
usage: timeline.py kernel_trace.csv [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bsx::", ""), r.get("Queue_Id", "?"),
              r.get("Stream_Id", "?")) for r in rows), key=lambda e: e[0])
# a step starts at k_header_merkle with the largest grid (main chunk); take the `back`-th last one
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_header_merkle")]
# group: consecutive merkle launches belong to the same step; find step heads = merkle whose previous merkle is > 3 ms earlier
heads = [starts[0]]
for a, b in zip(starts, starts[1:]):
    if ev[b][0] - ev[a][0] > 3_000_000:
        heads.append(b)
h = heads[-back]
t0 = ev[h][0]
nxt = heads[-back + 1] if back > 1 else len(ev)
for e in ev[h:nxt + 3]:
    print("%9.3f ms  +%8.3f ms  q%-3s s%-3s %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[4], e[2][:60]))
