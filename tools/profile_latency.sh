#!/bin/bash
# kernel timeline of ONE bsx_header_range call (host tier), from a kernel trace of tools/latency_probe.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/ktL -o lat -- python tools/latency_probe.py 30 > $O/ktL.log 2>&1
python tools/timeline3.py $(find $O/ktL -name "lat_kernel_trace.csv" | head -1) 1.1 0 > $O/r2_latency_timeline.txt
rm -rf $O/ktL
tail -2 $O/ktL.log; cat $O/r2_latency_timeline.txt
