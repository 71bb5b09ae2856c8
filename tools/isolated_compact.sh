#!/bin/bash
# tools: per-kernel durations of the compact step with NOTHING beside them (one buffer set, no commit check): isolated kernel times
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
d=gpurun_out/iso; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python bench.py --no-witness --engines 1 --alternate 1 --no-commit --no-autotune --steps 40 --no-legs --no-cpu-baseline >/dev/null 2>&1
python - "$(find $d -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("   %-44s n=%6s avg %9.1f us  total %8.2f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
