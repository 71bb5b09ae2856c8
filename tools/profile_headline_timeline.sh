#!/bin/bash
# steady-state kernel timeline of the headline pipeline (two chunks, witness materialised)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/ktH -o bench -- python bench.py --no-legs --steps 10 --warmup 2 > $O/ktH.log 2>&1
python tools/timeline2.py $(find $O/ktH -name "bench_kernel_trace.csv" | head -1) > $O/r2_timeline_steady_state.txt
rm -rf $O/ktH
cat $O/r2_timeline_steady_state.txt
