#!/usr/bin/env python3
"""Same-box A/B: step time with / without the commit verification on the side stream."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd.engine import HeaderRangeEngine
J, B, V, R = 32, 64, 100, 256
w = synth.Workload(4, R, J, B, v=V)
for rep in range(2):
    for commit in ("hash", "expand", False):
        if commit:
            os.environ["BSX_COMMIT_WITH"] = commit
        eng = HeaderRangeEngine(J, B, V, R, with_commit=bool(commit))
        if commit:
            eng.upload_workload(w)
        else:
            eng.upload(w.headers, w.ranges, w.latest)
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(15):
            eng.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 15
        print("commit", commit, "ms/step %.2f" % (dt * 1e3), "%.1f M/s" % (R * J * B / dt / 1e6))
        del eng
        torch.cuda.empty_cache()
