#!/usr/bin/env python3
"""Single-range latency of bsx_header_range (host tier) — for `rocprofv3 --kernel-trace --hip-trace --stats`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
J, B, V = 32, 64, 100
w = synth.Workload(4, 1, J, B, v=V)
f = InputDataFetcher(w.headers[0], int(w.first_height[0]), int(w.latest[0]))
circ = CombinedSkipCircuit(V, J, B)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    t0 = time.perf_counter(); circ.prove(w.input48(0), f, w.validators[0], w.trusted[0]); ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[3:]); print("median %.3f ms  min %.3f" % (ts[len(ts) // 2], ts[0]))
