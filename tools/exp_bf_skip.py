"""tools: VERDICT r5 #4 — what could ANY re-layout of the compact witness win for prove_subchain?  The experiments build's BSX_BF_SKIP
drops k_batch_finish's per-slot scattered stores (bit 0) and takes its strided predicate inputs from one broadcast line (bit 1): results
are wrong, the timing is the upper bound of a perfect layout.  Compact form of bench.py's compact_only leg (one chunk, two buffer sets),
no correctness gate.  usage (GPU box): BSX_LIB_OVERRIDE=$PWD/blobstreamx_amd/lib/libbsx_exp.so BSX_BF_SKIP={0,1,3} python tools/exp_bf_skip.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import synth
from blobstreamx_amd import _lib, engine as E
_lib.lib()
J, B, V, R = 32, 64, 100, 256
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
w = synth.Workload(4, R, J, B, v=V)
eng = E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=dev, with_witness=False)
eng.upload_workload(w)
for _ in range(5):
    eng.step()
eng.join(); torch.cuda.synchronize(dev)
eng.set_timing(True)
t0 = time.perf_counter()
N = 200
for _ in range(N):
    eng.step()
eng.join(); torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
tm = eng.timing()
print("BSX_BF_SKIP=%s: %.1f M headers/s  %.3f ms/step  prove_subchain %.4f ms/launch" % (os.environ.get("BSX_BF_SKIP", "0"), R * J * B * N / dt / 1e6, dt / N * 1e3, tm["prove_subchain_ms"]))
