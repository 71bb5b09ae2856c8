#!/usr/bin/env python3
"""Does the physical placement of the 29.5 GB witness buffer decide the expansion bandwidth?  Re-allocate ONLY that buffer
(holding on to the previous ones so the allocator must hand out different memory) and time the same expansion launch."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd.engine import HeaderRangeEngine
J, B, V, R = 32, 64, 100, 256
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R, with_commit=False)
eng.upload(w.headers, w.ranges, w.latest)
eng.step(); torch.cuda.synchronize()
L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
GB = eng.n_map_el * 8 / 1e9
def expand_ms(buf, n=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def once():
        _lib.check(L.bsx_dev_expand_witness(ctx, eng._st(), _lib.p(eng._ml), C.c_uint32(eng.RT * eng.jc), dp(eng.compact), dp(buf)))
    once(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
held = [eng.witness_map]
print("buffer 0 @%x: %.3f ms  %.0f GB/s" % (held[0].data_ptr(), expand_ms(held[0]), GB / expand_ms(held[0]) * 1e3), flush=True)
for i in range(1, 7):
    try:
        b = torch.empty(eng.n_map_el + 2, dtype=torch.int64, device="cuda")
    except RuntimeError as e:
        print("alloc failed", e); break
    held.append(b)
    t = expand_ms(b)
    print("buffer %d @%x: %.3f ms  %.0f GB/s" % (i, b.data_ptr(), t, GB / t * 1e3), flush=True)
# same buffers again (is the speed a property of the buffer?)
print("again:", " ".join("%.3f" % expand_ms(b) for b in held))
# free all but one, reallocate
keep = held[0]; held = None; torch.cuda.empty_cache()
for i in range(3):
    b = torch.empty(eng.n_map_el + 2, dtype=torch.int64, device="cuda")
    print("after free, realloc %d @%x: %.3f ms" % (i, b.data_ptr(), expand_ms(b)), flush=True)
    del b; torch.cuda.empty_cache()
