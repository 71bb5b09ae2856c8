"""Print the compact-only bench line in one row (tools; reads bench.py's JSON line on stdin)."""
import json, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_legs.line import detail_of
d = detail_of(sys.stdin.read())
cs = d.get("compact_step", {})
print("%.1f M headers/s  %.3f ms/step  frac %.3f  ceiling %.2f G/s  %s" % (
    d["value"] / 1e6, d["ms_per_step"], cs.get("frac_of_measured_alu_peak_whole_step", float("nan")),
    cs.get("measured_sha256_ceiling_per_s", 0) / 1e9,
    [(k.get("kernel", k.get("name", "?"))[:24], round(k["avg_launch_ms"], 3)) for k in d.get("kernels", []) if "avg_launch_ms" in k]))
