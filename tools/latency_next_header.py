#!/usr/bin/env python3
"""Wall time of one bsx_next_header call (CombinedStepCircuit through the host tier)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from blobstreamx_amd.builder import CombinedStepCircuit
V = 100
w = synth.Workload(4, 1, 1, 2, v=V, mode="S")
circ = CombinedStepCircuit(V)
prev, nxt = w.headers[0][0], w.headers[0][1]
inp = int(w.first_height[0]).to_bytes(8, "big") + w.hashes[0, 0].tobytes()
ts = []
for i in range(60):
    t0 = time.perf_counter()
    out, _ = circ.prove(inp, prev, nxt, int(w.latest[0]), w.validators[0][1] if w.validators[0].ndim == 2 else w.validators[0])
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[5:]); print("next_header: median %.3f ms  min %.3f" % (ts[len(ts) // 2], ts[0]))
