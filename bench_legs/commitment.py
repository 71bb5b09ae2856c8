"""fused_commitment: Poseidon Merkle caps of the map-job witnesses straight from the compact bytes."""
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
from .common import GL_MUL_PER_PERMUTATION, valu_issue


def commitment_leg(dev, J, B, V, cal, R=32, leaf_len=135, cap_height=4):
    """Poseidon (plonky2 PoseidonGoldilocksConfig) Merkle caps of every map job's witness.  (1) the pipeline's BSX_PIPE_CAPS
    mode: the whole step (hashing, hint, prove_subchain, reduce, commit check) + caps straight from the compact bytes, no
    64x image — headers/s, three un-joined steps, caps of two jobs checked against the oracle's own witness + Poseidon;
    (2) the commitment kernels alone: fused vs materialised (expand to HBM, then hash)."""
    import oracle
    import synth
    from blobstreamx_amd import _lib
    from blobstreamx_amd import engine as E
    from blobstreamx_amd.poseidon import WitnessCommitter
    w = synth.Workload(4, R, J, B, v=V)
    pe = E.PipelinedEngines(J, B, V, R, n_engines=2, device=dev, with_witness=False, with_caps=True, leaf_len=leaf_len, cap_height=cap_height)
    pe.upload_workload(w)
    pe.step()
    pe.join()
    pe.set_timing(True)
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        pe.step()
    pe.join()
    dt_pipe = (time.perf_counter() - t0) / steps
    tm = pe.timing()
    _, caps0 = pe.caps_numpy(0)
    res = pe.download()
    assert not res["range_status"].any() and not res["skip_status"].any()
    del pe
    eng = E.HeaderRangeEngine(J, B, V, R, device=dev)
    eng.upload_workload(w)
    eng.step()
    eng.join()
    n_jobs = R * J
    wc = WitnessCommitter(eng.ml, n_jobs, leaf_len, cap_height, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    compact, wmap = eng.compact, eng.witness_map

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize(dev)
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize(dev)
        return ev[0].elapsed_time(ev[1]) / reps
    t_fused = timed(lambda: wc.commit_compact(compact))
    caps_fused = wc.caps_numpy().copy()

    def materialised():
        _lib.check(eng.L.bsx_dev_expand_witness(eng.ctx, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), _lib.p(eng._ml),
                                                C.c_uint32(n_jobs), _lib.dp(compact), _lib.dp(wmap)))
        wc.commit_materialised(wmap)
    t_mat = timed(materialised)
    assert (wc.caps_numpy() == caps_fused).all(), "fused and materialised commitments differ"
    # the pipeline's chunk 0 holds ranges 0 .. R/2: its caps are the stand-alone committer's
    assert (caps0 == caps_fused[:caps0.shape[0]]).all(), "the pipeline's BSX_PIPE_CAPS output differs from the stand-alone commitment"
    # oracle check of two jobs: its own witness, its own Poseidon
    rc, _, _, cw = oracle.header_range(J, B, w.input48(0), w.headers[0], int(w.first_height[0]), int(w.latest[0]), w.validators[0],
                                       w.trusted[0], want_witness=True)
    full = oracle.expand_range_witness(J, B, cw)
    for j in (0, J - 1):
        _, cap = oracle.poseidon_merkle_tree(full[j * wc.nel:(j + 1) * wc.nel], leaf_len, wc.n_leaves, wc.cap_height)
        assert (caps_fused[j] == cap).all(), "witness commitment differs from the oracle"
    perms = n_jobs * wc.perms_per_job
    perm_per_s = perms / t_fused * 1e3
    peak = cal["goldilocks_mul_per_s"] / GL_MUL_PER_PERMUTATION
    # CPU leg = checker: the oracle's Poseidon over the SAME compact witnesses on all host threads (expand one job,
    # hash its rows, tree down to the cap), a bounded sample of the jobs, every cap compared with the GPU's
    cores, cores_desc = host_threads()
    compact_host = compact.cpu().numpy()
    n_cpu = min(n_jobs, 2 * cores)
    t0 = time.perf_counter()
    cpu_caps = oracle.bench_witness_caps(eng.ml, compact_host, n_cpu, leaf_len, wc.n_leaves, wc.cap_height, cores)
    dt_cpu = time.perf_counter() - t0
    creps = int(max(1, min(16, round(8.0 / max(dt_cpu, 1e-3)))))
    if creps > 1:
        t0 = time.perf_counter()
        cpu_caps = oracle.bench_witness_caps(eng.ml, compact_host, n_cpu, leaf_len, wc.n_leaves, wc.cap_height, cores, reps=creps)
        dt_cpu = time.perf_counter() - t0
    assert (cpu_caps == caps_fused[:n_cpu]).all(), "witness commitment: the CPU leg's caps differ from the GPU's"
    cpu_leg = {"value": n_cpu * creps * B / dt_cpu, "unit": "headers/s", "permutations_per_s": n_cpu * creps * wc.perms_per_job / dt_cpu,
               "cores": cores, "kind": "port",
               "sample": f"oracle Poseidon (128-bit accumulation form, oracle/poseidon.c) over the compact witnesses of {n_cpu} map jobs x {creps} "
                         f"repetitions: expand, hash {wc.n_leaves} rows of {leaf_len}, tree to the cap; {dt_cpu:.1f} s wall on {cores_desc}; "
                         f"all {n_cpu} caps equal the GPU's"}
    return {"workload": f"{R} x header_range_{J * B}: {n_jobs} map-job witnesses of {wc.nel} elements, rows of {leaf_len}, "
                        f"{wc.n_leaves} leaves, cap height {wc.cap_height}",
            "pipeline_caps_mode": {"headers_per_s": R * J * B / dt_pipe, "ms_per_step": dt_pipe * 1e3, "steps": steps, "caps_launch_ms": tm["caps_ms"],
                                   "note": "bsx_pipeline with BSX_PIPE_CAPS (no expansion): the whole step incl. commit check + Poseidon caps of every "
                                           "map-job witness from the compact bytes; steps not joined"},
            "fused_ms": t_fused, "materialised_ms": t_mat, "headers_per_s_fused": R * J * B / t_fused * 1e3,
            "permutations": perms, "checked_against_oracle_jobs": 2 + n_cpu, "cpu_baseline": cpu_leg,
            "roofline": {"kernel": "k_leaf_hashes<fused> + k_merkle_level", "bound": "valu", "unit": "G Poseidon permutations/s",
                         "achieved": perm_per_s / 1e9, "peak": peak / 1e9, "frac": min(1.0, perm_per_s / peak), "traffic": None,
                         "valu_issue": valu_issue(cal, "k_leaf_hashes<true>", n_jobs * wc.n_rows * (-(-leaf_len // 8)), t_fused * 1e-3),
                         "note": f"peak = the {GL_MUL_PER_PERMUTATION} Goldilocks multiplications of a permutation's x^7 S-boxes at the gl_mul rate measured in "
                                 f"this run ({cal['goldilocks_mul_per_s'] / 1e12:.2f} T/s) — the MDS layers' shift/add arithmetic is not counted, so this "
                                 "is an upper-bound style ceiling (an independent one: not the permutation's own micro-benchmark); "
                                 "valu_issue_frac = counted VALU wave-instructions / (time x measured v_add_u32 wave-issue rate)"},
            "hbm_bytes_not_written_per_header": int(8 * wc.nel / B)}
