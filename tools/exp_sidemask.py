#!/usr/bin/env python3
"""Commit side stream restricted to a CU subset (hipExtStreamCreateWithCUMask): same engine, interleaved rounds."""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth
from blobstreamx_amd.engine import HeaderRangeEngine

def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])

def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)

J, B, V, R = 32, 64, 100, 256
STEPS, ROUNDS = 30, 3
w = synth.Workload(4, R, J, B, v=V)
eng = HeaderRangeEngine(J, B, V, R)
eng.upload_workload(w)
hip = hip_runtime()
plain = eng.side
cfgs = {"nocommit": None, "plain": plain}
for n in (32, 64, 96, 128):
    cfgs["first%d" % n] = masked_stream(hip, range(n))
    cfgs["last%d" % n] = masked_stream(hip, range(256 - n, 256))
cfgs["every4th"] = masked_stream(hip, range(0, 256, 4))
res = {k: [] for k in cfgs}
for rnd in range(ROUNDS):
    for k, s in cfgs.items():
        eng.with_commit = s is not None
        if s is not None:
            eng.side = s
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(STEPS):
            eng.step()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t) / STEPS * 1e3)
for k in cfgs:
    print("%-10s" % k, " ".join("%.3f" % x for x in res[k]), flush=True)
