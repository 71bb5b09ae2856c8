"""tools: k_commit_tally alone (2048 commits x V validators, resident), HIP events: python tools/tally_bench.py [V] — with
BSX_LIB_OVERRIDE=.../libbsx_exp.so and BSX_TALLY_FORM=1 the workgroup forms (the A/B of round 6's wave form)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib, stress
V = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nh = 2048
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
L = _lib.lib(); ctx = _lib.context(0)
w = synth.CommitWorkload(5, nh, V) if hasattr(synth, "CommitWorkload") else None
sh = stress.CommitShard(nh, V, device=dev)
if w is None:
    ws = synth.Workload(5, 1, 32, 64, v=V)
    vals = np.repeat(ws.validators[:1], nh, axis=0); hh = np.repeat(ws.hashes[0, ws.n_blocks][None], nh, axis=0)
else:
    vals, hh = w.validators, w.hashes
sh.upload(vals, hh)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
dp = _lib.dp
for with_ok in (True, False):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for i in range(12):
        ev[0].record()
        _lib.check(L.bsx_dev_commit_tally(ctx, st, dp(sh.vals), C.c_uint32(sh.n), C.c_uint32(V), dp(sh.hh) if with_ok else None, dp(sh.ok) if with_ok else None, dp(sh.res)))
        ev[1].record(); torch.cuda.synchronize(dev)
        ts.append(ev[0].elapsed_time(ev[1]))
    P = 1 << max(0, (V - 1).bit_length())
    t = float(np.median(ts[2:]))
    print(f"V={V} with_ok={with_ok}: {t * 1e3:.1f} us  ({nh * (3 * P - 2) / t / 1e6:.2f} G compressions/s)  form {os.environ.get('BSX_TALLY_FORM', '0')}")
