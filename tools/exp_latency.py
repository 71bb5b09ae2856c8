"""tools (experiments build): where one bsx_header_range call spends its time — host enqueue vs wait for the GPU.
   usage on the GPU box:  BSX_LIB_OVERRIDE=$PWD/blobstreamx_amd/lib/libbsx_exp.so BSX_TRACE_HOST=1 python tools/exp_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import synth
from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher

J, B, V = 32, 64, 100
w = synth.Workload(4, 1, J, B, v=V)
f = InputDataFetcher(w.headers[0], int(w.first_height[0]), int(w.latest[0]), device=0)
circ = CombinedSkipCircuit(V, J, B, device=0)
ts = []
for i in range(40):
    t0 = time.perf_counter()
    o, _, _ = circ.prove(w.input48(0), f, w.validators[0], w.trusted[0])
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[5:])
print("pageable headers: median %.3f ms  min %.3f" % (ts[len(ts) // 2], ts[0]))
import torch
hp = torch.from_numpy(np.ascontiguousarray(w.headers[0]).view(np.uint8).reshape(-1)).pin_memory()
fp = InputDataFetcher(hp.numpy().view(w.headers.dtype), int(w.first_height[0]), int(w.latest[0]), device=0)
ts = []
for i in range(40):
    t0 = time.perf_counter()
    o2, _, _ = circ.prove(w.input48(0), fp, w.validators[0], w.trusted[0])
    ts.append((time.perf_counter() - t0) * 1e3)
assert o2 == o
ts = sorted(ts[5:])
print("page-locked headers: median %.3f ms  min %.3f" % (ts[len(ts) // 2], ts[0]))
