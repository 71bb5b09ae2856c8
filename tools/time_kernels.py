#!/usr/bin/env python3
"""Per-kernel wall time of one engine step (HIP events around each device-tier call), for kernel A/B experiments."""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd.engine import HeaderRangeEngine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
J, B, V = 32, 64, 100
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R, with_commit=False)
eng.upload_workload.__func__  # noqa
lo = eng.hfr
eng.upload(w.headers[:, lo:lo + eng.hpr], w.ranges, w.latest)
L, ctx, dp = eng.L, eng.ctx, _lib.dp
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
RT, jc, hpr = eng.RT, eng.jc, eng.hpr
calls = [
    ("header_merkle", lambda: L.bsx_dev_header_merkle(ctx, st, dp(eng.headers), C.c_uint64(RT * hpr), dp(eng.hashes), dp(eng.dh_aunts), dp(eng.lb_aunts), dp(eng.paths), dp(eng.status))),
    ("assemble", lambda: L.bsx_dev_assemble_inputs(ctx, st, C.c_uint32(RT), C.c_uint32(J), C.c_uint32(B), C.c_uint32(0), C.c_uint32(jc), C.c_uint32(B), dp(eng.ranges), dp(eng.latest), dp(eng.headers), C.c_uint64(hpr), C.c_uint64(0), dp(eng.hashes), dp(eng.dh_aunts), dp(eng.lb_aunts), dp(eng.compact), dp(eng.status[1:]), dp(eng.paths))),
    ("prove_subchain", lambda: L.bsx_dev_prove_subchain(ctx, st, C.c_uint32(RT), C.c_uint32(B), C.c_uint32(jc), dp(eng.ranges), dp(eng.compact), dp(eng.records), C.c_uint32(eng.subchain_flags))),
    ("reduce", lambda: L.bsx_dev_reduce(ctx, st, C.c_uint32(RT), C.c_uint32(jc), dp(eng.records), dp(eng.partial), dp(eng.red_compact_local))),
    ("expand", lambda: L.bsx_dev_expand_witness(ctx, st, _lib.p(eng._ml), C.c_uint32(RT * jc), dp(eng.compact), dp(eng.witness_map))),
]
tot = {k: 0.0 for k, _ in calls}
N = 6
for it in range(N + 2):
    for name, fn in calls:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(fn()); e1.record(); torch.cuda.synchronize()
        if it >= 2:
            tot[name] += e0.elapsed_time(e1) / N
print(os.environ.get("BSX_LIB_OVERRIDE", "default"), " ".join(f"{k}={v:.3f}ms" for k, v in tot.items()))
