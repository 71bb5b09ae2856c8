#!/usr/bin/env python3
"""Two full-size engines stepping alternately (software pipelining ACROSS steps) vs chunks of one step pipelined.
usage: exp_alternate.py [witness=0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, synth
from blobstreamx_amd.engine import PipelinedEngines
J, B, V, R = 32, 64, 100, 256
wit = len(sys.argv) > 1 and sys.argv[1] == "1"
w = synth.Workload(4, R, J, B, v=V)
for K, E in [tuple(int(c) for c in a.split(",")) for a in (sys.argv[2:] or ["1,2", "2,1", "1,1", "2,2"])]:
    engs = [PipelinedEngines(J, B, V, R, n_engines=E, with_witness=wit) for _ in range(K)]
    for e in engs:
        e.upload_workload(w)
        for _ in range(3): e.step()
        e.join()
    torch.cuda.synchronize()
    N = 40
    t0 = time.perf_counter()
    for i in range(N): engs[i % K].step()
    for e in engs: e.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("witness=%d K=%d E=%d  total %.3f ms/step  -> %.1f M headers/s" % (wit, K, E, (t2 - t0) / N * 1e3, R * J * B / ((t2 - t0) / N) / 1e6), flush=True)
    del engs; torch.cuda.empty_cache()
