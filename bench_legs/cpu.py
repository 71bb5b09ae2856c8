"""The `cpu_baseline` leg of bench.py: the C oracle (oracle/) timed on the host cores on a bounded sample, and used as the CHECKER of
the timed pipeline's outputs and witness.  This file and the `cpu_baseline` parts of stress.py / commitment.py are the only places of the
bench that import oracle/ — always after the GPU timing, never on the measured path."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import ROOT, HBM_PEAK_GBS, log, host_threads


def cpu_baseline_witness_check(eng, w, J, B, per_chunk=2):
    """Download the Goldilocks witness of `per_chunk` sampled ranges of every pipelined chunk — as the TIMED loop left it in
    HBM — and diff it, element by element, against the oracle's witness of the same range (map jobs of this rank's slice;
    at N = 1 also every reduce node), plus the public output of the owned ones.  Returns the number of ranges checked."""
    import oracle
    from blobstreamx_amd import engine as E
    from blobstreamx_amd import types as T
    ml, rl = T.map_layout(B), T.reduce_layout()
    nel, rel = int(ml["n_elements"]), int(rl["n_elements"])
    n = 0
    eng.join()
    out64 = eng.download()["output64"]
    for e in range(eng.E):
        sel = eng.sel(e)
        wm = eng.buffer(e, E.BUF_WITNESS_MAP, i64=True)
        wr = eng.buffer(e, E.BUF_WITNESS_REDUCE_LOCAL, i64=True)
        picks = sorted({0, eng.RT - 1} if per_chunk >= 2 else {0})
        for k in picks:
            r = int(sel[k])
            rc, out, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                                 w.validators[r], w.trusted[r], want_witness=True)
            assert rc == 0, f"oracle status {rc} on range {r}"
            full = oracle.expand_range_witness(J, B, cw)
            nm = eng.jc * nel
            got = wm[k * nm:(k + 1) * nm].cpu().numpy().view(np.uint64)
            want = full[eng.jf * nel:(eng.jf + eng.jc) * nel]
            assert (got == want).all(), f"map-job witness of range {r} differs from the oracle"
            if eng.world == 1 and J > 1:
                nr = (J - 1) * rel
                gr = wr[k * nr:(k + 1) * nr].cpu().numpy().view(np.uint64)
                assert (gr == full[J * nel:]).all(), f"reduce witness of range {r} differs from the oracle"
            own0 = eng.rank * eng.Rc
            if own0 <= k < own0 + eng.Rc:
                o = out64[e * eng.Rc + (k - own0)].tobytes()
                assert o == out, f"public output of range {r} differs from the oracle"
            n += 1
    return n


def cpu_baseline(w, J, B, V, seconds, gpu_out64, n_ranges, first=0):
    """Oracle (oracle/, C) timed on the host cores on a bounded sample of the SAME workload; its outputs double as a
    check of the GPU's public outputs for the sampled ranges."""
    import oracle
    cores, cores_desc = host_threads()
    n = n_ranges
    sl = slice(first, first + n)

    def run(reps):
        t = time.perf_counter()
        rc, out64, _ = oracle.bench_header_range(J, B, w.ranges[sl], w.headers[sl], w.hpr, w.latest[sl], w.validators[sl],
                                                 w.trusted[sl], V, True, cores, reps=reps)
        return time.perf_counter() - t, rc, out64
    r0 = max(1, -(-2 * cores // n))                 # >= 2 tasks per thread for the calibration pass
    dt, rc, out = run(r0)
    reps = int(max(r0, min(64 * r0, round(r0 * seconds / max(dt, 1e-3)))))
    if reps > r0:
        dt, rc, out = run(reps)
    assert rc == 0, f"oracle status {rc}"
    assert (out == gpu_out64[:n]).all(), "GPU public outputs differ from the oracle on the sampled ranges"
    return {"value": n * reps * J * B / dt, "unit": "headers/s", "cores": cores, "kind": "port",
            "sample": f"the {n} header_range_{J * B} instances of the GPU step x {reps} repetitions = {n * reps} ranges "
                      f"(same inputs, witness expansion included), {dt:.1f} s wall on {cores_desc}; outputs checked equal to the GPU's",
            "sha_ni": bool(oracle.has_shani()), "cpu_model": cores_desc}
