#!/usr/bin/env python3
"""Mode-S signature kernel at 2048 x V signatures by lanes per signature (experiments build: BSX_LIB_OVERRIDE=.../libbsx_exp.so,
BSX_ED_SPLIT = 1 / 2 / 4 read per process).  usage: exp_modeS_split.py V"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = synth.Workload(4 if V <= 100 else 5, 1, 32, 64, v=V, mode="S")
vals = np.ascontiguousarray(w.validators.reshape(-1)).view(np.uint8)
n = 2048 * V
dev = torch.device("cuda:0")
L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), dtype=torch.uint8, device=dev)
dv = torch.from_numpy(vals.copy()).to(dev)
dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
dok = torch.zeros(n, dtype=torch.uint8, device=dev)
scr = torch.zeros(int(L.bsx_ed25519_verify_scratch_bytes(C.c_uint64(n))), dtype=torch.uint8, device=dev)
_lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
_lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(V), dp(tab)))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for it in range(8):
    ev[0].record()
    _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(dok), dp(scr)))
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
assert int((dok == 1).sum().item()) == n
t = sorted(ts[2:])[len(ts[2:]) // 2]
print(f"V={V} n={n} BSX_ED_SPLIT={os.environ.get('BSX_ED_SPLIT', '-')} BSX_ED_BY_KEY={os.environ.get('BSX_ED_BY_KEY', '-')} keyed+finish {t:.3f} ms  {n / t / 1e3:.1f} M verifies/s")
