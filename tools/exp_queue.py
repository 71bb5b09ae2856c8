#!/usr/bin/env python3
"""Where does the run-to-run spread of the step time come from?  Re-create the engine N times (fresh buffers and a
fresh side stream each time), with and without the commit side stream, and re-time the SAME engine object."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd.engine import HeaderRangeEngine
J, B, V, R = 32, 64, 100, 256
STEPS = int(os.environ.get("STEPS", "40"))
w = synth.Workload(4, R, J, B, v=V)

def timed(eng):
    for _ in range(3):
        eng.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(STEPS):
        eng.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / STEPS * 1e3

for commit in (False, True):
    out = []
    for rep in range(6):
        eng = HeaderRangeEngine(J, B, V, R, with_commit=commit)
        if commit:
            eng.upload_workload(w)
        else:
            eng.upload(w.headers, w.ranges, w.latest)
        out.append("%.2f/%.2f/%.2f" % (timed(eng), timed(eng), timed(eng)))
        del eng
        torch.cuda.empty_cache()
    print("commit", commit, " ".join(out), flush=True)
