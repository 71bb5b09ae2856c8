"""tools: mode S (2048 x V) with one / two steps in flight (two CommitShard buffer sets on two streams)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import synth
from blobstreamx_amd.stress import CommitShard
V = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = synth.Workload(5, 1, 32, 64, v=V, mode="S")
vals = w.validators.reshape(2048, V)
shards = [CommitShard(2048, V) for _ in range(3)]
for s in shards:
    s.upload(vals, w.commit_hashes)
streams = [torch.cuda.Stream() for _ in range(3)]
for k in (1, 2, 3):
    for i in range(4): shards[i % k].step(streams[i % k])
    torch.cuda.synchronize()
    steps = 40
    t0 = time.perf_counter()
    for i in range(steps): shards[i % k].step(streams[i % k])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("V=%d  %d step(s) in flight: %.3f ms/step  %.2f M headers/s" % (V, k, dt * 1e3, 2048 / dt / 1e6))
