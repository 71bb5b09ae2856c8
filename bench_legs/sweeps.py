"""range_sweep (R resident ranges per step), keyset_churn (validator sets rotating between uploads), with_input_upload (PCIe-inclusive step)."""
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


def range_sweep_leg(dev, J, B, V, rs=(1, 4, 16, 64, 256), witness=False):
    """The pipeline's throughput by the number of ranges resident per step (R): how many concurrent proof requests it takes to fill
    the GPU.  Compact form (no expansion) unless `witness`; R = 1 .. 256, one chunk below 16 ranges, autotuned stream placement."""
    import synth
    from blobstreamx_amd import engine as E
    w = synth.Workload(4, max(rs), J, B, v=V)
    rows = []
    for R in rs:
        log(f"  range_sweep R={R} witness={witness}")
        nch = 2 if (witness and R >= 16) else 1
        pe = E.PipelinedEngines(J, B, V, R, n_engines=nch, device=dev, with_witness=witness) if witness else \
            E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=dev, with_witness=False)
        pe.upload_workload(w, sel=np.arange(R))
        pe.step()
        res = pe.download()
        assert not res["range_status"].any() and not res["skip_status"].any()
        steps = max(20, min(400, int(2000 // max(R, 1))))
        for _ in range(3):
            pe.step()
        pe.join()
        t0 = time.perf_counter()
        for _ in range(steps):
            pe.step()
        pe.join()
        dt = (time.perf_counter() - t0) / steps
        rows.append({"ranges": R, "headers_per_s": R * J * B / dt, "ms_per_step": dt * 1e3, "steps": steps})
        pe.close()
        del pe
        torch.cuda.empty_cache()
    return {"workload": f"bsx_pipeline_step over R resident header_range_{J * B} instances, {V} validators, "
                        + ("witness materialised" if witness else "compact form (no expansion), two buffer sets"), "by_ranges": rows}


def keyset_churn_leg(dev, J, B, V, R=64, ps=(0, 10, 100, 1000)):
    """Validator sets that CHANGE between the ranges of a chunk (VERDICT r4 missing #5; circuits/header_range.rs:42-48,
    circuits/fetcher.rs:60-87: `skip` exists because they do).  synth re-keys p / 1000 of the slots from range to range; the chunk's
    fixed-key Ed25519 table holds one row per DISTINCT public key (csrc/keycache.h), so the step's signature check stays on the table
    whatever p is (rounds 2-4: every slot whose key was not the first range's went to the generic kernel, 256 doublings).  Compact
    pipeline (two buffer sets) and the coalescing front end (16 native callers), every output checked against the chain's own hashes."""
    import synth
    from blobstreamx_amd import engine as E
    rows = []
    for p in ps:
        w = synth.Workload(6, R, J, B, v=V, rotate_permille=p)
        distinct = len({bytes(k) for r in range(R) for k in w.validators[r]["pubkey"]})
        pe = E.AlternatingPipelines(2, J, B, V, R, n_engines=1, device=dev, with_witness=False)
        t0 = time.perf_counter()
        pe.upload_workload(w)
        t_up = time.perf_counter() - t0
        pe.step()
        res = pe.download()
        assert not res["range_status"].any() and not res["skip_status"].any(), (p, res["skip_status"])
        for r in range(R):
            assert res["output64"][r][:32].tobytes() == w.hashes[r, w.n_blocks].tobytes()
        for _ in range(3):
            pe.step()
        pe.join()
        steps = 60
        t0 = time.perf_counter()
        for _ in range(steps):
            pe.step()
        pe.join()
        dt = (time.perf_counter() - t0) / steps
        rows.append({"rotate_permille": p, "distinct_keys": distinct, "table_MB": distinct * 5.8, "ms_per_step": dt * 1e3, "headers_per_s": R * J * B / dt,
                     "upload_s_incl_table_build": t_up})
        pe.close()
        del pe
        torch.cuda.empty_cache()
    base = rows[0]["ms_per_step"]
    for r in rows:
        r["step_time_vs_p0"] = r["ms_per_step"] / base
    return {"workload": f"bsx_pipeline_step, compact form, {R} resident header_range_{J * B} instances, {V} validators, validator set of range r + 1 = "
                        "range r's with p / 1000 of its slots re-keyed", "by_rotate_permille": rows,
            "note": "table rows are keyed by public key and built at bsx_pipeline_upload (where validator sets change); a step launches no table work"}


def upload_leg(eng, args, steps, tune_streams=True):
    """The headline step with the header block (headers + skip headers, 512 B each) streamed from pinned host memory EVERY
    step on a copy stream inside the library (bsx_pipeline_enable_input_streaming), overlapped with the previous step's
    compute: the PCIe-inclusive rate of a caller whose inputs are not resident."""
    from blobstreamx_amd import engine as E
    eng.enable_input_streaming(True)
    tune = eng.autotune(0) if tune_streams else None        # the copy streams take queues too: place the chunks' streams for THIS mode
    for _ in range(2):
        eng.step()
    eng.join()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    eng.join()
    dt = (time.perf_counter() - t0) / steps
    eng.enable_input_streaming(False)
    nbytes = sum(eng.buffer(e, E.BUF_HEADERS).numel() for e in range(eng.E))
    return {"value": eng.R * args.jobs * args.batch / dt, "unit": "headers/s", "ms_per_step": dt * 1e3, "steps": steps,
            "h2d_bytes_per_step": nbytes, "h2d_GBps": nbytes / dt / 1e9,
            "stream_autotune": tune,
            "note": "inputs streamed H2D from pinned memory on a copy stream each step, overlapped with compute; the witness stays on the device"}
