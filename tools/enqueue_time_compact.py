#!/usr/bin/env python3
"""Host-side cost of enqueueing one COMPACT-ONLY step (no expansion) vs its GPU time, by number of pipelined chunks and with /
without the commit check (which runs on side streams).  usage: enqueue_time_compact.py [E ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from blobstreamx_amd.engine import PipelinedEngines
J, B, V, R = 32, 64, 100, 256
w = synth.Workload(4, R, J, B, v=V)
Es = [int(x) for x in sys.argv[1:]] or [2, 1, 4]
for commit in (True, False):
    for E in Es:
        eng = PipelinedEngines(J, B, V, R, n_engines=E, with_witness=False, with_commit=commit)
        eng.upload_workload(w)
        for _ in range(5): eng.step()
        eng.join(); torch.cuda.synchronize()
        N = 40
        t0 = time.perf_counter()
        for _ in range(N): eng.step()
        t1 = time.perf_counter()
        eng.join(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("E=%d commit=%d  host enqueue %.3f ms/step   total %.3f ms/step  -> %.1f M headers/s" % (E, commit, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, R * J * B / ((t2 - t0) / N) / 1e6), flush=True)
        del eng; torch.cuda.empty_cache()
