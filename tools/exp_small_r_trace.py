import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import synth
from blobstreamx_amd import engine as E
J, B, V = 32, 64, 100
R = int(sys.argv[1])
w = synth.Workload(4, R, J, B, v=V)
pe = E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=torch.device("cuda:0"), with_witness=False)
pe.upload_workload(w, sel=np.arange(R))
for _ in range(6):
    pe.step(); pe.join()
