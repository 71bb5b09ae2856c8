"""Single-call latency, K concurrent callers (own contexts / coalesced through the batcher) and the 32-hint burst of one proof."""
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


def latency_leg(dev, J, B, V):
    """One proof request: bsx_header_range (host tier: host pointers in, 64 B out [+ witness]) for ONE header_range, median of
    50 calls.  The host tier takes its device scratch from a per-context arena (no hipMalloc per call)."""
    import synth
    from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
    w = synth.Workload(4, 1, J, B, v=V)
    f = InputDataFetcher(w.headers[0], int(w.first_height[0]), int(w.latest[0]), device=dev.index or 0)
    circ = CombinedSkipCircuit(V, J, B, device=dev.index or 0)
    out = {}
    for key, ww, n in (("output_only_ms", False, 50), ("with_witness_download_ms", True, 20)):
        ts = []
        for i in range(n + 3):
            t0 = time.perf_counter()
            o, _, _ = circ.prove(w.input48(0), f, w.validators[0], w.trusted[0], want_witness=ww)
            ts.append((time.perf_counter() - t0) * 1e3)
        assert o[:32] == w.hashes[0, w.n_blocks].tobytes()
        ts = sorted(ts[3:])
        out[key] = {"median": ts[len(ts) // 2], "min": ts[0], "p90": ts[int(len(ts) * 0.9)], "calls": n}
    # the other entry point of the reference (bin/next_header.rs: CombinedStepCircuit, one header + its V-validator commit): one
    # bsx_next_header call, 40 B in, 64 B out; its output is checked against the oracle in tests/test_gpu_units.py
    from blobstreamx_amd.builder import CombinedStepCircuit
    ws = synth.Workload(4, 1, 1, 2, v=V, mode="S")
    step = CombinedStepCircuit(V, device=dev.index or 0)
    inp40 = int(ws.first_height[0]).to_bytes(8, "big") + ws.hashes[0, 0].tobytes()
    vals1 = ws.validators[0][1] if ws.validators[0].ndim == 2 else ws.validators[0]
    ts = []
    for i in range(43):
        t0 = time.perf_counter()
        o40, _ = step.prove(inp40, ws.headers[0][0], ws.headers[0][1], int(ws.latest[0]), vals1)
        ts.append((time.perf_counter() - t0) * 1e3)
    assert o40[:32] == ws.hashes[0, 1].tobytes()
    ts = sorted(ts[3:])
    out["next_header_ms"] = {"median": ts[len(ts) // 2], "min": ts[0], "p90": ts[int(len(ts) * 0.9)], "calls": 40,
                             "workload": f"one bsx_next_header (CombinedStepCircuit): 2 headers + a {V}-validator commit, host pointers in, 64 B out"}
    out["workload"] = f"one header_range_{J * B}, {V} validators, through bsx_header_range (H2D of {J * B + 1} headers + validators, all kernels, D2H)"
    out["headers_per_s_single_stream"] = J * B / out["output_only_ms"]["median"] * 1e3
    return out


def _concdrive():
    """tests/hostcheck/libconcdrive.so: K NATIVE threads calling the host tier (Python threads spend ~20 us under the GIL per ctypes
    call — at 50,000 calls/s that is the whole budget, and the library would not be what is measured).  Built by build()."""
    from blobstreamx_amd import _lib
    _lib.lib()                                                   # libbsx.so first: the driver binds to the same copy
    D = C.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libconcdrive.so"))
    D.cd_header_range_loop.restype = C.c_double
    D.cd_hint_burst.restype = C.c_int
    return D


def concurrent_leg(dev, J, B, V, ks=(1, 2, 4, 8, 16, 32, 64), seconds=0.5, window_us=0, n_lanes=0, max_requests=0, pinned=False, serial=True,
                   forms=("pageable", "page_locked", "registered", "packed"), form_ks=(1, 16, 64)):
    """The reference's shape of use: ONE range per `prove` call, several calls in flight under a multi-thread runtime
    (circuits/header_range.rs:180-181, bin/header_range_2048.rs:6-17).  K native threads call the UNCHANGED bsx_header_range back to
    back, each on its own range (host pointers in, 64 B out), for `seconds`:
      coalesced   all threads share ONE context with bsx_enable_coalescing: calls arriving together run as one launch set.  Four upload
                  forms (round 6): `pageable` host memory (staged into the batcher's page-locked block by the caller's thread), `page_locked`
                  (hipHostMalloc'ed: uploaded from where it lies), `registered` (the same pageable buffers after ONE bsx_host_register each:
                  what a host that reuses its buffer does), `packed` (bsx_header_range_packed: ~408 instead of 512 bytes per header)
      serial      every thread has its OWN context and the calls run one by one per context (round 4's shape), K = 1 and 16
    headers/s, p50 / p99 per call, requests per launch set; every output checked against the chain's own target hash."""
    import synth
    from blobstreamx_amd import _lib
    from blobstreamx_amd import batcher as BT
    L = _lib.lib()
    D = _concdrive()
    D.cd_header_range_loop2.restype = C.c_double
    kmax = max(ks)
    n_w = min(kmax, 32)                                          # distinct ranges (threads beyond take them again)
    w = synth.Workload(4, n_w, J, B, v=V)
    cid = np.frombuffer(b"celestia", np.uint8).copy()
    inp = np.stack([np.frombuffer(w.input48(k % n_w), np.uint8) for k in range(kmax)]).copy()
    hdrs = [np.ascontiguousarray(w.headers[k % n_w]).copy() for k in range(kmax)]
    keep = [torch.empty(h.nbytes, dtype=torch.uint8, pin_memory=True) for h in hdrs] if (pinned or "page_locked" in forms) else []
    for t_, h in zip(keep, hdrs):
        t_.numpy()[:] = h.view(np.uint8).reshape(-1)
    hdrs_pinned = [t_.numpy().view(hdrs[0].dtype) for t_ in keep]
    packed = [BT.pack_headers(h) for h in hdrs] if "packed" in forms else []
    tv = [np.ascontiguousarray(w.validators[k % n_w]) for k in range(kmax)]
    rv = [np.ascontiguousarray(w.trusted[k % n_w]) for k in range(kmax)]
    fh = np.array([int(w.first_height[k % n_w]) for k in range(kmax)], np.uint64)
    lt = np.array([int(w.latest[k % n_w]) for k in range(kmax)], np.uint64)
    PP = C.c_void_p * kmax
    p_tv, p_rv = PP(*[x.ctypes.data for x in tv]), PP(*[x.ctypes.data for x in rv])
    cap = 1 << 16

    def run(ctx_handles, shared, K, bufs=None, form=0):
        bufs = hdrs if bufs is None else bufs
        p_h = PP(*[h.ctypes.data for h in bufs])
        nh = np.array([bufs[k].size for k in range(kmax)], np.uint64)      # headers, or BYTES of a packed block
        lat = np.zeros((K, cap), np.float32)
        counts, rcs, o64 = np.zeros(K, np.int32), np.zeros(K, np.int32), np.zeros((K, 64), np.uint8)
        CT = C.c_void_p * len(ctx_handles)
        dt = D.cd_header_range_loop2(CT(*[h.value for h in ctx_handles]), C.c_int(shared), C.c_int(K), C.c_double(seconds), C.c_uint32(J), C.c_uint32(B),
                                     C.c_uint32(V), _lib.p(inp), p_h, _lib.p(fh), _lib.p(nh), _lib.p(lt), p_tv, p_rv, _lib.p(cid), C.c_uint32(8), _lib.p(lat),
                                     C.c_int(cap), _lib.p(counts), _lib.p(o64), _lib.p(rcs), C.c_int(form))
        assert dt > 0 and not rcs.any(), (dt, rcs)
        for k in range(K):
            assert o64[k, :32].tobytes() == w.hashes[k % n_w, w.n_blocks].tobytes()
        allv = np.sort(np.concatenate([lat[k, :min(cap, counts[k])] for k in range(K)]))
        n = int(counts.sum())
        per_call = sum(int(bufs[k].nbytes) for k in range(K)) / K
        return {"threads": K, "calls": n, "headers_per_s": n * J * B / dt, "calls_per_s": n / dt, "p50_ms": float(allv[len(allv) // 2]),
                "p99_ms": float(allv[min(len(allv) - 1, int(len(allv) * 0.99))]), "h2d_GBps_implied": n * per_call / dt / 1e9}

    # coalesced: one shared context
    shared_ctx = C.c_void_p()
    _lib.check(L.bsx_init(C.c_int(dev.index or 0), C.byref(shared_ctx)))
    cfg = BT.make_config(J, B, V, window_us=window_us, n_lanes=n_lanes, max_requests=max_requests)
    _lib.check(L.bsx_enable_coalescing(shared_ctx, C.byref(cfg)))
    L.bsx_context_batcher.restype = C.c_void_p
    view = BT.Batcher(J, B, V, handle=C.c_void_p(L.bsx_context_batcher(shared_ctx)))
    run([shared_ctx], 1, min(8, kmax))                           # warm: lanes, key tables

    def sweep(Ks, bufs=None, form=0):
        rows = []
        for K in Ks:
            s0 = view.stats()["header_range"]
            row = run([shared_ctx], 1, K, bufs, form)
            s1 = view.stats()["header_range"]
            nb = max(1, s1["batches"] - s0["batches"])
            row["requests_per_launch_set"] = (s1["requests"] - s0["requests"]) / nb
            # the worker's time per launch set by phase (us): collecting, staging (+ enqueuing the header uploads), enqueuing the kernels, waiting
            # for the GPU, publishing + taking out the results nobody has taken yet
            row["worker_us_per_set"] = {k: round((s1[k] * s1["batches"] - s0[k] * s0["batches"]) / nb, 1)
                                        for k in ("close_wait_us", "stage_wait_us", "enqueue_us", "gpu_wait_us", "complete_us")}
            rows.append(row)
        return rows
    out_forms = {}
    rows = sweep(ks, hdrs_pinned if pinned else None)
    fks = [k for k in form_ks if k <= kmax]
    if "page_locked" in forms and not pinned:
        out_forms["coalesced_page_locked"] = sweep(fks, hdrs_pinned)
    if "packed" in forms:
        out_forms["coalesced_packed_headers"] = sweep(fks, packed, form=1)
    if "registered" in forms and not pinned:
        for h in hdrs:                                           # ONE registration per buffer, reused by every call from it
            _lib.check(L.bsx_host_register(shared_ctx, C.c_void_p(h.ctypes.data), C.c_uint64(h.nbytes)))
        try:
            out_forms["coalesced_registered_once"] = sweep(fks)
        finally:
            for h in hdrs:
                L.bsx_host_unregister(shared_ctx, C.c_void_p(h.ctypes.data))
    view.h = None
    L.bsx_shutdown(shared_ctx)
    # serial: own contexts (round 4's shape)
    serial_rows = []
    ctxs = []
    for _ in range(min(16, kmax) if serial else 0):
        h = C.c_void_p()
        _lib.check(L.bsx_init(C.c_int(dev.index or 0), C.byref(h)))
        ctxs.append(h)
    if ctxs:
        run(ctxs, 0, len(ctxs))                                  # warm every context
        for K in (1, len(ctxs)):
            serial_rows.append(run(ctxs[:K], 0, K))
    for h in ctxs:
        L.bsx_shutdown(h)
    return {"workload": f"K native threads x bsx_header_range (one header_range_{J * B}, {V} validators per call, host pointers in, 64 B out)",
            "coalesced_shared_context": rows, **out_forms, "serial_own_contexts": serial_rows, "headers_page_locked": bool(pinned),
            "bytes_per_call": {"records": int(hdrs[0].nbytes), "packed": int(packed[0].nbytes) if packed else None},
            "pcie_note": f"every call uploads {(J * B + 1) * 512 / 1e6:.2f} MB of header records ({(packed[0].nbytes if packed else 0) / 1e6:.2f} MB packed): 100 M headers/s = "
                         "51 GB/s of H2D, the PCIe Gen5 x16 practical ceiling (with_input_upload measures ~46 GB/s on these boxes) — h2d_GBps_implied says how close a row is",
            "note": "the reference proves ONE range per call under a multi-thread runtime (header_range.rs:180-181): this is that shape.  coalesced_shared_context "
                    "= pageable caller memory (the default row); the other coalesced_* rows change only how the headers reach the GPU"}


def hint_concurrent_leg(dev, J, B, V, reps=40):
    """The map-job hints of ONE proof issued the way the reference issues them: one `async fn hint` per map job under the runtime
    (circuits/builder.rs:325-332 -> circuits/data_commitment.rs:22-44), each followed by prove_subchain (builder.rs:335).  J native
    threads, thread j calls the UNCHANGED bsx_data_commitment_inputs (65 headers in, 2 x 64 proofs out) [+ bsx_prove_subchain] for map
    job j; all released together; wall time from release to the last return, median / p90 of `reps` bursts — on a context with
    coalescing enabled and on a plain one (calls take turns on the context's lock: round 4).  Records checked against the oracle."""
    import oracle
    import synth
    from blobstreamx_amd import _lib
    from blobstreamx_amd import batcher as BT
    from blobstreamx_amd import types as T
    L = _lib.lib()
    D = _concdrive()
    w = synth.Workload(4, 1, J, B, v=V)
    S, latest, E = int(w.first_height[0]), int(w.latest[0]), int(w.first_height[0]) + J * B
    hdr = np.ascontiguousarray(w.headers[0])
    end_hash = np.ascontiguousarray(w.hashes[0, J * B])
    out = {"workload": f"the {J} map-job hints of one header_range_{J * B} from {J} native threads (bsx_data_commitment_inputs: {B + 1} headers in, "
                       f"{B} + {B} inclusion proofs out; then bsx_prove_subchain), {reps} bursts"}
    want = None
    for mode in ("coalesced", "serial"):
        ctx = C.c_void_p()
        _lib.check(L.bsx_init(C.c_int(dev.index or 0), C.byref(ctx)))
        if mode == "coalesced":
            cfg = BT.make_config(J, B, V)
            _lib.check(L.bsx_enable_coalescing(ctx, C.byref(cfg)))
        res = {}
        for key, sub in (("hint_only", 0), ("hint_then_prove_subchain", 1), ("map_job_one_call", 2)):
            wall = np.zeros(reps + 5, np.float32)
            recs = np.zeros(J, T.SUBCHAIN)
            se = np.zeros((J, 64), np.uint8)
            rc = D.cd_hint_burst(ctx, C.c_int(J), C.c_uint32(B), C.c_int(reps + 5), C.c_int(sub), _lib.p(hdr), C.c_uint64(S), C.c_uint64(latest),
                                 C.c_uint64(E), _lib.p(end_hash), _lib.p(wall), _lib.p(recs), _lib.p(se))
            assert rc == 0, rc
            ws = np.sort(wall[5:])
            res[key] = {"median_ms": float(ws[len(ws) // 2]), "p90_ms": float(ws[int(len(ws) * 0.9)]), "min_ms": float(ws[0])}
            if sub:                                              # checker: the oracle's records for the same map jobs
                if want is None:
                    want = []
                    for j in range(J):
                        bs, be = S + j * B, S + (j + 1) * B
                        _, oh = oracle.data_commitment_inputs(hdr[j * B:(j + 1) * B + 1], bs, latest, bs, be, B)
                        want.append(oracle.prove_subchain(B, oh["start_header"], oh["end_header"], oh["data_hash_proofs"], oh["last_block_id_proofs"], bs, be,
                                                          E, end_hash.tobytes())[1])
                for j in range(J):
                    assert recs[j].tobytes() == want[j].tobytes(), (mode, j)
        if mode == "coalesced":
            L.bsx_context_batcher.restype = C.c_void_p
            st = BT.Batcher(J, B, V, handle=C.c_void_p(L.bsx_context_batcher(ctx))).stats()
            res["requests_per_launch_set"] = {k: st[k]["requests"] / max(1, st[k]["batches"]) for k in ("data_commitment_inputs", "prove_subchain")}
            res["worker_us_per_set"] = {k: {q: round(st[k][q], 1) for q in ("close_wait_us", "stage_wait_us", "enqueue_us", "gpu_wait_us", "complete_us")}
                                        for k in ("data_commitment_inputs", "prove_subchain")}
            res["note"] = "map_job_one_call = bsx_map_job: the map closure (builder.rs:305-336: hint, then prove_subchain) as ONE coalesced request"

        out[mode] = res
        L.bsx_shutdown(ctx)
    out["records_checked_vs_oracle"] = J
    return out
