#!/bin/bash
# tools: A/B of two builds of the library on the compact-only step, with per-kernel averages (rocprofv3 --kernel-trace --stats).
#   usage (on the GPU box): bash tools/ab_compact.sh <libA.so> <libB.so>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  for i in 1 2; do
    BSX_LIB_OVERRIDE=$PWD/$lib timeout 300 python bench.py --no-witness --engines 1 --alternate 2 --steps 200 --no-legs --no-cpu-baseline 2>/dev/null | python tools/compact_line.py
  done
  d=gpurun_out/ab/$(basename $lib .so); rm -rf $d; mkdir -p $d
  BSX_LIB_OVERRIDE=$PWD/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python bench.py --no-witness --engines 1 --alternate 2 --steps 40 --no-legs --no-cpu-baseline >/dev/null 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("   %-44s n=%6s avg %9.1f us  total %8.2f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
