"""tools: compact pipeline step time at few ranges, by the number of un-joined steps in the loop."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import synth
from blobstreamx_amd import engine as E
J, B, V = 32, 64, 100
w = synth.Workload(4, 16, J, B, v=V)
for R in (1, 4, 8, 16):
    pe = E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=torch.device("cuda:0"), with_witness=False)
    pe.upload_workload(w, sel=np.arange(R))
    pe.step(); pe.join()
    out = []
    for steps in (1, 2, 20, 100, 400):
        for _ in range(3): pe.step()
        pe.join()
        t0 = time.perf_counter()
        for _ in range(steps): pe.step()
        t1 = time.perf_counter()
        pe.join()
        dt = (time.perf_counter() - t0) / steps
        out.append("%d steps: %.3f ms/step (enqueue %.3f)" % (steps, dt * 1e3, (t1 - t0) / steps * 1e3))
    print("R=%d  " % R + " | ".join(out))
    pe.close(); del pe
