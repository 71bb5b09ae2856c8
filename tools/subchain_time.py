#!/usr/bin/env python3
"""prove_subchain alone on the GPU at the bench shape (524,288 slots in one launch): the one-launch form and the separate-launch
form, over the compact witnesses a real pipeline step left.  usage: subchain_time.py [ranges=256]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import engine as E
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
J, B, V = 32, 64, 100
w = synth.Workload(4, R, J, B, v=V)
eng = E.HeaderRangeEngine(J, B, V, R, with_witness=False, with_commit=False)
eng.upload_workload(w)
eng.step(); eng.join()
L, ctx, dp = eng.L, eng.ctx, _lib.dp
compact, records, ranges = eng.buffer(0, E.BUF_COMPACT), eng.buffer(0, E.BUF_RECORDS), eng.buffer(0, E.BUF_RANGES)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, flags in (("one launch (k_batch_finish<fused>)", 1), ("separate launches", 3)):
    fn = lambda: _lib.check(L.bsx_dev_prove_subchain(ctx, st, C.c_uint32(R), C.c_uint32(B), C.c_uint32(J), dp(ranges), dp(compact), dp(records), C.c_uint32(flags)))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per {R * J * B} slots")
