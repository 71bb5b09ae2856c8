"""tools (experiments build, BSX_ABLATE): the compact step's time with the hint and / or prove_subchain left out (timing only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import synth
from blobstreamx_amd import engine as E
J, B, V, R = 32, 64, 100, 256
w = synth.Workload(4, R, J, B, v=V)
pe = E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=torch.device("cuda:0"), with_witness=False, with_commit=os.environ.get("COMMIT", "1") == "1")
pe.upload_workload(w)
pe.autotune(0)
for _ in range(10): pe.step()
pe.join()
t0 = time.perf_counter()
for _ in range(200): pe.step()
pe.join()
print("BSX_ABLATE=%s commit=%s: %.3f ms/step" % (os.environ.get("BSX_ABLATE", "0"), os.environ.get("COMMIT", "1"), (time.perf_counter() - t0) / 200 * 1e3))
