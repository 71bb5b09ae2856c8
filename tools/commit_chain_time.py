#!/usr/bin/env python3
"""Stage-by-stage GPU time of the mode-F commit check (device tier, one stream, HIP events) at pipelined-chunk sizes:
challenge, decode R, key-table check, signature check (inversion form vs latency form), tallies, and the whole chain."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import synth
from blobstreamx_amd import _lib

L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
dev = torch.device("cuda:0")
V = 100
for R in (1, 128, 256):
    w = synth.Workload(4, R, 2, 4, v=V)
    vals = w.validators.reshape(-1)
    n = vals.size
    dv = torch.from_numpy(vals.view(np.uint8).copy()).to(dev)
    dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    rd = torch.zeros(int(L.bsx_ed25519_decoded_r_bytes(C.c_uint64(n))), dtype=torch.uint8, device=dev)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    hh = torch.from_numpy(w.hashes[:, w.n_blocks].copy()).to(dev).view(-1)
    res = torch.zeros(R * 96, dtype=torch.uint8, device=dev)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    stages = [
        ("sha512_challenge", lambda: L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None)),
        ("decode_r", lambda: L.bsx_dev_ed25519_decode_r(ctx, st, dp(dv), C.c_uint64(n), dp(rd))),
        ("keytable", lambda: L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(V), dp(tab))),
        ("verify_keyed (inversion form)", lambda: L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(ok), None)),
        ("verify_keyed_r (latency form)", lambda: L.bsx_dev_ed25519_verify_keyed_r(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(rd), dp(ok))),
        ("commit_tally", lambda: L.bsx_dev_commit_tally(ctx, st, dp(dv), C.c_uint32(R), C.c_uint32(V), dp(hh), dp(ok), dp(res))),
    ]
    print(f"--- {R} commits x {V} = {n} signatures")
    for name, fn in stages:
        for _ in range(2):
            _lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            _lib.check(fn())
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:34s} {e0.elapsed_time(e1) / reps * 1e3:9.1f} us")
    assert int(ok.sum()) == n
