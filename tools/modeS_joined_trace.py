#!/usr/bin/env python3
"""Joined mode-S steps (one in flight) for a kernel trace: 2048 x V commits, `n` steps of CommitShard.step() + gather().
usage: rocprofv3 --kernel-trace --output-format csv -d out -o ms -- python tools/modeS_joined_trace.py [V=100] [n=12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd.stress import CommitShard

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w = synth.Workload(4, 1, 32, 64, v=V, mode="S")
sh = CommitShard(2048, V, n_sets=1)
sh.upload(w.validators.reshape(2048, V), w.commit_hashes)
for _ in range(3):
    sh.gather(sh.step())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    sh.gather(sh.step())
torch.cuda.synchronize()
print("joined step: %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
