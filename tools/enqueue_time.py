#!/usr/bin/env python3
"""Host-side cost of enqueueing one step (ctypes + torch stream/event calls) vs the GPU time of the step."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from blobstreamx_amd.engine import PipelinedEngines
for B in (64, 32):
    J, V, R = 32, 100, 256
    w = synth.Workload(4, R, J, B, v=V)
    eng = PipelinedEngines(J, B, V, R, n_engines=2)
    eng.upload_workload(w)
    for _ in range(5): eng.step()
    eng.join(); torch.cuda.synchronize()
    N = 40
    t0 = time.perf_counter()
    for _ in range(N): eng.step()
    t1 = time.perf_counter()
    eng.join(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d  host enqueue %.3f ms/step   total %.3f ms/step" % (B, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3), flush=True)
    del eng; torch.cuda.empty_cache()
