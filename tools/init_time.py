#!/usr/bin/env python3
"""Wall time of bsx_init (context creation: streams, zero-path constants, the 64 MB Ed25519 table of B)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blobstreamx_amd import _lib
torch.cuda.init(); torch.zeros(1, device="cuda")
L = _lib.lib()
for i in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.bsx_init(C.c_int(0), C.byref(h))
    t1 = time.perf_counter()
    print("bsx_init rc=%d  %.2f ms" % (rc, (t1 - t0) * 1e3))
    L.bsx_shutdown(h)
