#!/usr/bin/env python3
"""The 32 hints of one proof from 32 native threads on a coalescing context, a few bursts, for a rocprofv3 trace
(rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -o hb -- python tools/hint_burst_trace.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, synth
from blobstreamx_amd import _lib, batcher as BT, types as T

J, B, V = 32, 64, 100
L = _lib.lib()
D = bench._concdrive()
w = synth.Workload(4, 1, J, B, v=V)
S, latest, E = int(w.first_height[0]), int(w.latest[0]), int(w.first_height[0]) + J * B
hdr = np.ascontiguousarray(w.headers[0]); end_hash = np.ascontiguousarray(w.hashes[0, J * B])
ctx = C.c_void_p(); _lib.check(L.bsx_init(C.c_int(0), C.byref(ctx)))
cfg = BT.make_config(J, B, V); _lib.check(L.bsx_enable_coalescing(ctx, C.byref(cfg)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wall = np.zeros(reps, np.float32); recs = np.zeros(J, T.SUBCHAIN); se = np.zeros((J, 64), np.uint8)
rc = D.cd_hint_burst(ctx, C.c_int(J), C.c_uint32(B), C.c_int(reps), C.c_int(0), _lib.p(hdr), C.c_uint64(S), C.c_uint64(latest), C.c_uint64(E), _lib.p(end_hash),
                     _lib.p(wall), _lib.p(recs), _lib.p(se))
print("rc", rc, "burst ms:", " ".join("%.3f" % x for x in wall))
