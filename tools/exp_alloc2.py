#!/usr/bin/env python3
"""The REAL expansion launch (4096 map jobs, 14.7 GB) on buffers from different allocators, isolated: hipMalloc (torch) x N held
simultaneously vs bsx_dev_alloc (VMM) x N held simultaneously, in the order given on the command line."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BSX_WITNESS_ALLOC"] = "torch"
import numpy as np, torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd.engine import HeaderRangeEngine

J, B, V, R = 32, 64, 100, 128
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R, with_witness=False)
eng.upload_workload(w); eng.step(); torch.cuda.synchronize()
n_jobs, nel = R * J, int(eng.ml["n_elements"])
n_el = n_jobs * nel + 2
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
job_bytes = int(eng.ml["n_bytes"]) + 4 * int(eng.ml["n_words"]) + int(eng.ml["n_bools"]) + 8 * nel

def t_ms(buf):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        if it == 1: ev[0].record()
        _lib.check(eng.L.bsx_dev_expand_witness(eng.ctx, st, _lib.p(eng._ml), C.c_uint32(n_jobs), _lib.dp(eng.compact), _lib.dp(buf)))
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 3

held = []
for kind in (sys.argv[1:] or ["torch"] * 5 + ["vmm"] * 5 + ["torch"] * 3):
    if kind == "vmm":
        blk = _lib.DeviceBuffer(n_el); buf = blk.tensor(); held.append(blk)
    else:
        buf = torch.zeros(n_el, dtype=torch.int64, device="cuda")
    held.append(buf)
    ms = t_ms(buf)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    buf.fill_(1); ev[0].record()
    for _ in range(3): buf.fill_(2)
    ev[1].record(); torch.cuda.synchronize()
    fill = ev[0].elapsed_time(ev[1]) / 3
    print(f"{kind:5s} ptr {buf.data_ptr():#x}  expand {ms:.3f} ms  {n_jobs * job_bytes / ms / 1e6:.0f} GB/s   torch.fill_ {fill:.3f} ms {n_el * 8 / fill / 1e6:.0f} GB/s", flush=True)
print("re-time all:", " ".join(f"{t_ms(b):.3f}" for b in held if isinstance(b, torch.Tensor)))
