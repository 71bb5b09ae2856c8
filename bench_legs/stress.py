"""Mode S (a commit on every header: BASELINE config #5) — Ed25519 + SHA-512 bound; its own `roofline` (bound "valu") and `cpu_baseline`."""
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
from .common import FE_SQ_PER_VERIFY, fe_mul_per_verify, keyed_verify_peak, valu_issue, pmc_traffic


def stress(args, dev, V, cpu_seconds, cal, rank=0, world=1, check=True):
    """Mode S (BASELINE configs #4/#5: 'N headers x V validators', i.e. next_header.rs:25-47 per header): every header of one
    header_range_2048 carries its own V-signature commit.  Rank g verifies commits [g*N/world, (g+1)*N/world) through ONE
    C-ABI call per step (bsx_dev_verify_commits) and ONE all-gather of the 128-byte folds tells every rank the verdict of
    the whole range.  Per-signature ok bits, every commit result and the fold of this rank's slice are compared with the
    oracle's (the CPU leg); the gathered folds are compared with the oracle's folds of every slice on rank 0."""
    import synth
    from blobstreamx_amd import _lib
    from blobstreamx_amd.stress import CommitShard, range_verdict
    nh = args.jobs * args.batch
    w = synth.Workload(5 if V > 100 else 4, 1, args.jobs, args.batch, v=V, mode="S")
    # three buffer sets on three streams: steps i + 1, i + 2 start while step i's stages drain (stress.py CommitShard; 2048 x 100:
    # 1.03 / 0.68 / 0.61 ms per step with 1 / 2 / 3 in flight); the fold all-gather of step i is taken while they run
    wide = None if getattr(args, "wide_tables", "auto") == "auto" else args.wide_tables == "1"
    sh = CommitShard(nh, V, rank=rank, world=world, device=dev, n_sets=3, wide_tables=wide)
    sh.upload(w.validators.reshape(nh, V), w.commit_hashes)
    n = sh.n * V
    L, ctx, dp = sh.L, sh.ctx, _lib.dp
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    h_scratch = sh.scratch           # challenge scalars live at the head of the scratch block (bsx.h)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()

    def staged():
        """the stages of bsx_dev_verify_commits as separate device-tier calls, bracketed by HIP events on the launch stream"""
        ed_scr = sh.scratch[((n * 32 + 255) & ~255):]
        ev[0].record()
        _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(sh.vals), C.c_uint64(n), dp(h_scratch), None))
        ev[1].record()
        _lib.check(L.bsx_dev_ed25519_keytable_w(ctx, st, dp(sh.vals), C.c_uint32(V), dp(sh.keytable), C.c_uint32(sh.kt_bits)))
        ev[2].record()
        _lib.check(L.bsx_dev_ed25519_verify_keyed_w(ctx, st, dp(sh.vals), dp(h_scratch), C.c_uint64(n), C.c_uint32(V), dp(sh.keytable),
                                                    C.c_uint32(V), dp(sh.ok), dp(ed_scr), C.c_uint32(sh.kt_bits)))
        ev[3].record()
        _lib.check(L.bsx_dev_commit_tally(ctx, st, dp(sh.vals), C.c_uint32(sh.n), C.c_uint32(V), dp(sh.hh), dp(sh.ok), dp(sh.res)))
        ev[4].record()
        torch.cuda.synchronize(dev)
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    cold = staged()                                 # first call: every table row is built
    t = np.mean([staged() for _ in range(5)], axis=0)
    t_sha, t_tab, t_ed, t_tally = (float(x) for x in t)
    # the timed object: K steps of the ONE composite call + the fold all-gather, barrier on both sides, max over ranks
    def timed(shard, K, in_flight):
        """K steps with `in_flight` of them enqueued at any time: the folds of step i are gathered (the collective + a host sync on
        that step's stream) once step i + in_flight - 1 has been enqueued; in_flight = 1 is the joined loop"""
        for _ in range(shard.K):                      # every buffer set once (first touch of its pages, its stream's first launch)
            shard.gather(shard.step())
        barrier()
        t0 = time.perf_counter()
        pending, folds = [], None
        for _ in range(K):
            pending.append(shard.step())
            if len(pending) >= in_flight:
                folds = shard.gather(pending.pop(0))
        while pending:
            folds = shard.gather(pending.pop(0))
        barrier()
        dt = (time.perf_counter() - t0) / K
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, folds
    dt1, _ = timed(sh, 5, 1)
    dt, folds = timed(sh, 30, sh.K)
    tot = dt * 1e3
    gpu_ok, gpu_res, gpu_fold = sh.download()
    verdict = range_verdict(folds)
    out = {"workload": f"mode S: {nh} headers x {V} validators = {nh * V} signatures (one header_range_{nh}, a commit per header)"
                       + (f", sharded {world} x {sh.n} commits, one all-gather of 128-byte folds" if world > 1 else ""),
           "headers_per_s": nh / dt, "verifies_per_s_all_stages": nh * V / dt, "ms": tot, "steps_in_flight": sh.K,
           "one_step_in_flight": {"headers_per_s": nh / dt1, "ms": dt1 * 1e3},
           "verifies_per_s_incl_table": n / (t_ed + t_tab) * 1e3, "signatures": nh * V, "signatures_this_rank": n,
           "range_verdict": verdict,
           "stage_ms": {"sha512_challenge": t_sha, "keytable": t_tab, "ed25519_verify_keyed": t_ed, "tally_validator_hash": t_tally,
                        "keytable_cold_build": cold[1]},
           "keytable_digit_bits": sh.kt_bits, "keytable_MB": round(sh.keytable.numel() / 1e6, 1),
           "ed25519_path": ("fixed-key affine tables: %s of h for every validator key, 16 radix-65536 digits of s for B (64 MB) = %d mixed additions, no "
                            "doubling; table rows reused while the validator set is unchanged; encodings through per-lane Montgomery batch inversion "
                            "(8 / 16 / 32 signatures per inversion)") % (("16 radix-65536 digits (64 MB per key: BSX_COMMITS_KEYTABLE_WIDE, the validator set "
                            "is resident)", 32) if sh.kt_bits == 16 else ("22 radix-4096 digits (5.8 MB per key)", 38))}
    ver_per_s = n / (t_ed * 1e-3)
    peak = keyed_verify_peak(cal, sh.kt_bits)
    out["roofline"] = {"kernel": "k_ed25519_verify_keyed", "bound": "valu", "unit": "M Ed25519 verifications/s",
                       "achieved": ver_per_s / 1e6, "peak": peak / 1e6, "frac": min(1.0, ver_per_s / peak),
                       "avg_launch_ms": t_ed, "traffic": None,
                       "field_ops_per_verification": {"mul": fe_mul_per_verify(sh.kt_bits), "sq": FE_SQ_PER_VERIFY},
                       "achieved_G_field_ops_per_s": ver_per_s * (fe_mul_per_verify(sh.kt_bits) + FE_SQ_PER_VERIFY) / 1e9,
                       # 2048 x 100 runs k_ed25519_verify_keyed_mixed (kernels_ed.hip: whole waves per SIMD one lane per signature, the rest on four)
                       "valu_issue": valu_issue(cal, "k_ed25519_verify_keyed_mixed" if n < 300000 else "k_ed25519_verify_keyed<true, true, 1>", n, t_ed * 1e-3),
                       "note": "peak = the time the kernel's GF(2^255-19) multiplications and squarings would take at the fe_mul / fe_sq "
                               f"rates measured in this run ({cal['fe25519_mul_per_s'] / 1e9:.0f} / {cal['fe25519_sq_per_s'] / 1e9:.0f} G/s); additions, "
                               "table selection, recoding and the launch's partial last wave round are what is left; ALU bound, bytes are "
                               "not the limiter (96 B in per signature)",
                       "sha512_challenge": {"avg_launch_ms": t_sha, "compressions_per_s": 2 * n / t_sha * 1e3,
                                            "frac_of_measured_peak": min(1.0, 2 * n / t_sha * 1e3 / cal["sha512_compress_per_s"]),
                                            "algorithmic_GBps": n * 237 / t_sha / 1e6},
                       # the validator-set trees: P = V rounded up to a power of two leaves (one compression each) + P - 1 inner nodes (two)
                       "commit_tally": (lambda P: {"avg_launch_ms": t_tally, "sha256_compressions": sh.n * (3 * P - 2),
                                                   "compressions_per_s": sh.n * (3 * P - 2) / t_tally * 1e3,
                                                   "frac_of_measured_peak": min(1.0, sh.n * (3 * P - 2) / t_tally * 1e3 / cal["sha256_compress_per_s"]),
                                                   "note": "a latency chain: 1 + 2 log2 P dependent compressions per commit (50 us at the one-wave rate); "
                                                           "with BSX_COMMITS_TALLY_BESIDE (the timed steps) it runs beside the signature check"})(1 << max(0, (V - 1).bit_length()))}
    if not check:
        return out
    import oracle
    # The WITNESS of the per-validator loop (BASELINE config #5: "bit-exact witness diff vs CPU"): the same call also leaves every
    # commit's COMPACT COMMIT unit (digests, challenges, verdicts, leaves, the masked validator-set tree, sums: include/bsx_layout.h),
    # expanded into Goldilocks elements by k_expand_witness on the same stream — HBM-write bound, its own roofline next to the VALU one.
    del sh
    torch.cuda.empty_cache()
    shw = CommitShard(nh, V, rank=rank, world=world, device=dev, expand=True, n_sets=2, wide_tables=wide)
    shw.upload(w.validators.reshape(nh, V), w.commit_hashes)
    lay = shw.lay
    exp_bytes = shw.n * (int(lay["n_bytes"]) + 4 * int(lay["n_words"]) + int(lay["n_bools"]) + 8 * int(lay["n_elements"]))
    evw = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    layp = np.ascontiguousarray(lay).reshape(1)
    shw.step(); shw.step()
    torch.cuda.synchronize(dev)                     # the steps ran on the sets' own streams
    t_x = 0.0
    for _ in range(3):
        evw[0].record()
        _lib.check(L.bsx_dev_expand_witness(ctx, st, _lib.p(layp), C.c_uint32(shw.n), dp(shw.compact), dp(shw.witness)))
        evw[1].record()
        torch.cuda.synchronize(dev)
        t_x += evw[0].elapsed_time(evw[1]) / 3
    dtw1, _ = timed(shw, 3, 1)
    dtw, _ = timed(shw, 8, shw.K)
    pick = sorted({0, 1, shw.n // 3, shw.n // 2, shw.n - 1})
    got = shw.witness_of(pick)
    vv_all = w.validators.reshape(nh, V)
    for i, c in enumerate(pick):
        gc = shw.first + c
        _, _, cwc = oracle.verify_commit(vv_all[gc], w.commit_hashes[gc].tobytes(), want_witness=True)
        want = oracle.expand_witness(lay, 1, cwc)
        assert got[i].shape == want.shape and (got[i] == want).all(), f"mode S: the COMMIT unit of commit {gc} differs from the oracle's"
    gpu_ok_w, gpu_res_w, gpu_fold_w = shw.download()
    unit_traffic, unit_traffic_src = pmc_traffic(shw.n, {"layout": "commit", "v": V})
    out["witness"] = {"headers_per_s": nh / dtw, "ms": dtw * 1e3, "steps_in_flight": shw.K,
                      "one_step_in_flight": {"headers_per_s": nh / dtw1, "ms": dtw1 * 1e3}, "elements_per_commit": int(lay["n_elements"]),
                      "bytes_per_step_this_rank": int(shw.n * 8 * int(lay["n_elements"])),
                      "checked_against_oracle_commits": len(pick),
                      "roofline": {"kernel": "k_expand_witness (COMMIT units)", "bound": "hbm", "achieved": exp_bytes / t_x / 1e6, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": exp_bytes / t_x / 1e6 / HBM_PEAK_GBS, "avg_launch_ms": t_x,
                                   "algorithmic_bytes_per_launch": exp_bytes, "traffic": unit_traffic, "traffic_source": unit_traffic_src,
                                   "frac_of_measured_store_ceiling": min(1.0, exp_bytes / t_x * 1e3 / cal["hbm_store_bytes_per_s"])},
                      "note": "the whole mode-S step WITH the witness: verification + compact units + their 64x expansion into HBM; sampled commits' "
                              "units diffed element by element against the oracle"}
    assert (gpu_ok_w == gpu_ok).all() and gpu_fold_w.tobytes() == gpu_fold.tobytes(), "mode S: emitting the witness changed the verdicts"
    sh = shw
    # CPU leg = checker: the oracle's verify_commit of this rank's commits on all host threads, repeated to fill ~cpu_seconds
    cores, cores_desc = host_threads()
    vv = w.validators.reshape(nh, V)[sh.first:sh.first + sh.n]
    hh = w.commit_hashes[sh.first:sh.first + sh.n]
    t0 = time.perf_counter()
    res, ok = oracle.bench_verify_commits(vv, hh, cores, reps=1)
    dtc = time.perf_counter() - t0
    creps = int(max(1, min(32, round(cpu_seconds / max(dtc, 1e-3)))))
    if creps > 1:
        t0 = time.perf_counter()
        res, ok = oracle.bench_verify_commits(vv, hh, cores, reps=creps)
        dtc = time.perf_counter() - t0
    assert (gpu_ok == ok).all(), "mode S: per-signature verdicts differ from the oracle"
    a, b = gpu_res.copy(), res.copy()
    a["_pad"] = 0; b["_pad"] = 0
    if a.tobytes() != b.tobytes():
        bad = [c for c in range(sh.n) if a[c].tobytes() != b[c].tobytes()]
        raise AssertionError(f"mode S: commit results differ from the oracle at {len(bad)} commits, first {bad[:4]}: {a[bad[0]]} vs {b[bad[0]]}")
    assert int(gpu_ok.sum()) == n
    ofold = oracle.commit_fold(res, sh.first)
    assert gpu_fold.tobytes() == ofold.tobytes(), "mode S: this rank's fold differs from the oracle's fold of the oracle's results"
    assert folds[rank].tobytes() == ofold.tobytes(), "mode S: the gathered fold of this rank is not the one it sent"
    assert verdict["all_ok"] and verdict["commits"] == nh, verdict
    out["checked_against_oracle"] = {"sig_ok_bits": n, "commit_results": sh.n, "fold": 1, "gathered_folds": int(len(folds))}
    out["cpu_baseline"] = {"value": sh.n * creps / dtc, "unit": "headers/s", "verifies_per_s": n * creps / dtc, "cores": cores, "kind": "port",
                           "sample": f"oracle verify_commit of this rank's {sh.n} commits x {creps} repetitions, {dtc:.1f} s wall on {cores_desc}; "
                                     "every verdict, commit result and the fold compared with the GPU's"}
    return out
