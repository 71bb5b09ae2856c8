#!/bin/bash
# compact-only pipeline (no Goldilocks expansion): kernel trace of `bench.py --no-witness`; the ALU-bound form of the path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktC -o bench -- python bench.py --no-legs --no-witness --engines 1 --steps 10 --warmup 2 > $O/ktC.log 2>&1
python tools/kernel_avg.py $(find $O/ktC -name "bench_kernel_trace.csv" | head -1) > $O/r2_compact_kernel_avg_steady_state.txt
python tools/timeline3.py $(find $O/ktC -name "bench_kernel_trace.csv" | head -1) 7 1 > $O/r2_compact_timeline.txt
rm -rf $O/ktC
tail -1 $O/ktC.log | cut -c1-400; cat $O/r2_compact_kernel_avg_steady_state.txt
