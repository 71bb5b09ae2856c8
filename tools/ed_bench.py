#!/usr/bin/env python3
"""Ed25519 / SHA-512 kernel throughput at scale (mode S building blocks): n signatures = 25,600 unique x reps."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib

def main():
    reps_list = [int(x) for x in (sys.argv[1:] or ["1", "4", "16"])]
    w = synth.Workload(4, 1, 4, 64, v=100, mode="S")
    base = w.validators.reshape(-1).view(np.uint8).reshape(-1, 256)
    dev = torch.device("cuda:0")
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for reps in reps_list:
        n = base.shape[0] * reps
        dv = torch.from_numpy(np.tile(base, (reps, 1)).copy()).to(dev)
        dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        dok = torch.zeros(n, dtype=torch.uint8, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for it in range(3):
            ev[0].record()
            _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
            ev[1].record()
            _lib.check(L.bsx_dev_ed25519_verify(ctx, st, dp(dv), dp(dh), C.c_uint64(n), dp(dok)))
            ev[2].record()
            torch.cuda.synchronize()
        assert int(dok.sum().item()) == n
        t1, t2 = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        print(f"n={n:8d}  sha512 {t1:8.3f} ms ({n/t1/1e3:7.1f} M/s, {n*237/t1/1e6:7.1f} GB/s)   ed25519 {t2:8.3f} ms ({n/t2/1e3:7.2f} M verifies/s)")

if __name__ == "__main__":
    main()
