#!/usr/bin/env python3
"""Same-box, same-buffers A/B: step time by commit placement (BSX_COMMIT_WITH) and P7 form (BSX_ED_PATH).
ONE engine object, so every configuration runs on the same allocations; configurations are interleaved over rounds."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["BSX_ED_PATH"] = "keyed"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd.engine import HeaderRangeEngine
J, B, V, R = 32, 64, 100, 256
STEPS = int(os.environ.get("STEPS", "30"))
ROUNDS = int(os.environ.get("ROUNDS", "4"))
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R)
eng.upload_workload(w)
cfgs = [(False, "generic"), ("hash", "generic"), ("hash", "keyed"), ("expand", "generic"), ("expand", "keyed")]
res = {c: [] for c in cfgs}
for rnd in range(ROUNDS):
    for c in cfgs:
        eng.with_commit = bool(c[0])
        eng.commit_with = c[0] or "hash"
        eng.ed_path = c[1]
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(STEPS):
            eng.step()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t) / STEPS * 1e3)
for c in cfgs:
    print("%-8s %-8s" % c, " ".join("%.3f" % x for x in res[c]), " min %.3f" % min(res[c]), flush=True)
