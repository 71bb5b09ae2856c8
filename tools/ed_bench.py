#!/usr/bin/env python3
"""Ed25519 / SHA-512 kernel throughput at scale (mode S building blocks): n signatures = 25,600 unique x reps,
generic per-signature path and fixed-key (per-validator table) path side by side."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib

def main():
    reps_list = [int(x) for x in (sys.argv[1:] or ["1", "4", "16"])]
    V = 100
    w = synth.Workload(4, 1, 4, 64, v=V, mode="S")
    base = w.validators.reshape(-1).view(np.uint8).reshape(-1, 256)
    dev = torch.device("cuda:0")
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), dtype=torch.uint8, device=dev)
    for reps in reps_list:
        n = base.shape[0] * reps
        dv = torch.from_numpy(np.tile(base, (reps, 1)).copy()).to(dev)
        dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        dok = torch.zeros(n, dtype=torch.uint8, device=dev)
        dok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
        scr = torch.zeros(n * 160, dtype=torch.uint8, device=dev) if os.environ.get('ED_BATCH_INV', '1') == '1' else None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        for it in range(3):
            ev[0].record()
            _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
            ev[1].record()
            _lib.check(L.bsx_dev_ed25519_verify(ctx, st, dp(dv), dp(dh), C.c_uint64(n), dp(dok)))
            ev[2].record()
            _lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(V), dp(tab)))
            ev[3].record()
            _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(dok2), dp(scr) if scr is not None else None))
            ev[4].record()
            torch.cuda.synchronize()
        assert int(dok.sum().item()) == n and int(dok2.sum().item()) == n
        t1, t2, t3, t4 = (ev[i].elapsed_time(ev[i + 1]) for i in range(4))
        print(f"n={n:8d}  sha512 {t1:8.3f} ms ({n/t1/1e3:7.1f} M/s)   ed25519 generic {t2:8.3f} ms ({n/t2/1e3:7.2f} M/s)   "
              f"keytable {t3:7.3f} ms + keyed {t4:8.3f} ms ({n/(t3+t4)/1e3:7.2f} M/s incl. table)", flush=True)

if __name__ == "__main__":
    main()
