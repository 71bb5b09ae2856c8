#!/usr/bin/env python3
"""Where the wall time of one bsx_header_range call goes: Python wrapper vs the C call (H2D, ~17 launches, D2H, sync)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth
from blobstreamx_amd import _lib, types as T
from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
J, B, V = 32, 64, 100
w = synth.Workload(4, 1, J, B, v=V)
f = InputDataFetcher(w.headers[0], int(w.first_height[0]), int(w.latest[0]))
circ = CombinedSkipCircuit(V, J, B)
inp = w.input48(0)
for _ in range(5): circ.prove(inp, f, w.validators[0], w.trusted[0])
tv = np.ascontiguousarray(w.validators[0], T.VALIDATOR).reshape(-1); rv = np.ascontiguousarray(w.trusted[0], T.VALIDATOR).reshape(-1)
out = np.zeros(64, np.uint8); res = np.zeros(1, T.COMMIT_RESULT)
L, ctx = _lib.lib(), _lib.context(0)
cid = np.frombuffer(b"celestia", np.uint8).copy()
inb = np.frombuffer(inp, np.uint8).copy()
args = (ctx, C.c_uint32(J), C.c_uint32(B), _lib.p(inb), _lib.p(f.headers), C.c_uint64(f.first_height), C.c_uint64(f.headers.size),
        C.c_uint64(f.latest_block), _lib.p(tv), _lib.p(rv), C.c_uint32(V), _lib.p(cid), C.c_uint32(8), _lib.p(out), _lib.p(res), None)
ts, tp = [], []
for i in range(100):
    t0 = time.perf_counter(); rc = L.bsx_header_range(*args); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); circ.prove(inp, f, w.validators[0], w.trusted[0]); tp.append(time.perf_counter() - t0)
assert rc == 0
ts.sort(); tp.sort()
print("C call alone: median %.3f ms (min %.3f)   prove() wrapper: median %.3f ms" % (ts[50] * 1e3, ts[0] * 1e3, tp[50] * 1e3))
