#!/usr/bin/env python3
"""CU-mask partition experiment: can the HBM-bound witness expansion saturate HBM from a subset of the CUs while the
ALU-bound hashing runs on the rest?  Streams created with hipExtStreamCreateWithCUMask, passed through the C ABI."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd.engine import HeaderRangeEngine

def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not mapped")

def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

J, B, V, R = 32, 64, 100, 256
w = synth.Workload(4, R, J, B, v=V)
engs = [HeaderRangeEngine(J, B, V, R, with_commit=False) for _ in range(2)]
for e in engs:
    e.upload(w.headers, w.ranges, w.latest)
    e.step()
torch.cuda.synchronize()
hip = hip_runtime()
L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp

def expand(e):
    _lib.check(L.bsx_dev_expand_witness(ctx, e._st(), _lib.p(e._ml), C.c_uint32(e.RT * e.jc), dp(e.compact), dp(e.witness_map)))

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

GB = engs[0].n_map_el * 8 / 1e9
print("witness GB per expansion %.2f" % GB)
patterns = {
    "first": lambda n: list(range(n)),
    "stride": lambda n: [i for i in range(256) if (i * n) // 256 != ((i - 1) * n) // 256 or i == 0][:n],
}
for name, pat in patterns.items():
    for n in (256, 192, 128, 96, 64, 32):
        bits = pat(n)
        sm = masked_stream(hip, bits)
        rest = [i for i in range(256) if i not in set(bits)]
        def f_exp():
            with torch.cuda.stream(sm):
                expand(engs[0])
        te = timeit(f_exp)
        line = "%-6s expand on %3d CUs: %.3f ms (%.0f GB/s)" % (name, len(bits), te, GB / te * 1e3)
        if rest:
            sa = masked_stream(hip, rest)
            def f_hash():
                with torch.cuda.stream(sa):
                    engs[1].step_local()
            th = timeit(f_hash)
            def f_both():
                with torch.cuda.stream(sm):
                    expand(engs[0])
                with torch.cuda.stream(sa):
                    engs[1].step_local()
            tb = timeit(f_both)
            line += "   hashing on %3d CUs: %.3f ms   both concurrently: %.3f ms" % (len(rest), th, tb)
        print(line, flush=True)
# reference: unmasked, two plain streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def f_both_plain():
    with torch.cuda.stream(s1):
        expand(engs[0])
    with torch.cuda.stream(s2):
        engs[1].step_local()
def f_seq():
    expand(engs[0]); engs[1].step_local()
print("plain streams concurrently: %.3f ms   sequential on one stream: %.3f ms" % (timeit(f_both_plain), timeit(f_seq)))
