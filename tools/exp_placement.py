#!/usr/bin/env python3
"""Does the physical placement of the witness buffer decide the expansion bandwidth?  Re-allocate ONLY that buffer
(holding on to the previous ones so the allocator must hand out different memory) and time the same expansion launch;
then the same with hipExtMallocWithFlags(hipDeviceMallocContiguous)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd.engine import HeaderRangeEngine
os.environ["BSX_PLACEMENT_PROBE"] = "1"
J, B, V, R = 32, 64, 100, int(os.environ.get("R", "128"))
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R, with_commit=False)
eng.upload(w.headers, w.ranges, w.latest)
eng.step(); torch.cuda.synchronize()
L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
GB = eng.n_map_el * 8 / 1e9
def expand_ms(ptr, n=6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def once():
        _lib.check(L.bsx_dev_expand_witness(ctx, eng._st(), _lib.p(eng._ml), C.c_uint32(eng.RT * eng.jc), dp(eng.compact), ptr))
    once(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("witness GB %.2f" % GB)
held = [eng.witness_map]
for i in range(1, 6):
    held.append(torch.empty(eng.n_map_el + 2, dtype=torch.int64, device="cuda"))
print("default hipMalloc :", " ".join("%.3f" % expand_ms(dp(b)) for b in held), "ms", flush=True)
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1]); break
nbytes = (eng.n_map_el + 2) * 8
ptrs = []
for flag, name in ((0x4, "contiguous"), (0x3, "uncached"), (0x1, "finegrained")):
    out = []
    for i in range(3):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flag))
        if rc != 0:
            out.append("alloc rc=%d" % rc); break
        ptrs.append(p)
        out.append("%.3f" % expand_ms(p))
    print("%-12s      :" % name, " ".join(out), flush=True)
