#!/usr/bin/env python3
"""bench.py — headers/sec of header_range witness generation on MI355X (BASELINE.json metric).

A "step" = one pass of the whole hot path (header hashing + hint assembly + prove_subchain + reduce + final
asserts + target-commit verification + Goldilocks witness expansion) over one batch of R synthetic header_range_2048
instances (32 map jobs x 64 headers, 100 validators, mode F = one commit per range, exactly what one reference proof
does) whose inputs are already resident in HBM.  value = N * R * 2048 headers / step time, all ranks, max over ranks.

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): every rank computes its 32/N-job slice of all
N*R ranges, one all-gather of 128-byte records per pipelined chunk, the owner finishes its R ranges — weak scaling
(blobstreamx_amd/engine.py).

Extra objects on the JSON line: `roofline` (dominant kernel = witness expansion, HBM-write bound), `kernels` (the SHA
kernels' compact-byte rates, never mixed with the expanded figure), `cpu_baseline` (the C oracle timed on this box's
host cores on a bounded sample, N = 1 only), `stress` (mode S: one 100-signature commit per header, Ed25519 bound).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Two pipelined chunks need four independent hardware queues (2 main + 2 commit side streams) beside the default stream,
# the library's own stream and — at N > 1 — RCCL's; HIP's default of 4 maps the second chunk's main stream onto the first
# chunk's side-stream queue and serialises them.  8 and 16 measure the same at N = 1 (91.8 / 91.3 M headers/s); 16 leaves
# headroom for the collective's streams.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ranges", type=int, default=256, help="header_range instances per GPU per step (R)")
    ap.add_argument("--jobs", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--validators", type=int, default=100)
    ap.add_argument("--engines", type=int, default=2, help="chunks of the step pipelined on separate HIP streams (the ALU-bound hashing "
                    "of one chunk beside the HBM-bound expansion of the other; phase tokens keep the chunks in complementary "
                    "phases).  Measured on MI355X with GPU_MAX_HW_QUEUES=8: 72.8 / 80.5-81.3 / ~52 M headers/s at 1 / 2 / 4 chunks "
                    "(4 chunks: the co-running chunks starve each other; more hardware queues do not help: 67 M at 16 queues).  At N > 1 every chunk does its own all-gather")
    ap.add_argument("--event-every", type=int, default=1, help="record the per-kernel HIP events on every n-th timed step")
    ap.add_argument("--no-witness", action="store_true", help="skip the Goldilocks expansion (reported as such)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(w, J, B, V, seconds, gpu_out64):
    """Oracle (oracle/, C) timed on the host cores on a bounded sample of the SAME workload; its outputs double as a
    check of the GPU's public outputs for the sampled ranges."""
    import oracle
    cores = os.cpu_count() or 1
    n = w.R

    def run(reps):
        t = time.perf_counter()
        rc, out64, _ = oracle.bench_header_range(J, B, w.ranges[:n], w.headers[:n], w.hpr, w.latest[:n], w.validators[:n],
                                                 w.trusted[:n], V, True, cores, reps=reps)
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
                      f"(same inputs, witness expansion included), {dt:.1f} s wall on {cores} threads; outputs checked equal to the GPU's",
            "sha_ni": bool(oracle.has_shani())}


def stress(args, dev):
    """Mode S (BASELINE configs' 'N headers x V validators'): every header of one 2048-range carries its own
    100-signature commit: 204,800 Ed25519 verifications + SHA-512 challenges + 2048 validator-set hashes per range."""
    import ctypes as C
    import synth
    from blobstreamx_amd import _lib
    nh, V = args.jobs * args.batch, args.validators       # one whole header_range_2048 with a commit on every header
    w = synth.Workload(4, 1, args.jobs, args.batch, v=V, mode="S")
    vals = w.validators.reshape(-1)
    n = vals.size
    L, ctx, dp = _lib.lib(), _lib.context(dev.index or 0), _lib.dp
    dv = torch.from_numpy(vals.view(np.uint8).copy()).to(dev)
    dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    dhh = torch.from_numpy(w.commit_hashes.copy()).to(dev)
    dres = torch.zeros(nh * 96, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    def once():
        ev[0].record()
        _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
        ev[1].record()
        # fixed-key P7: the per-validator tables are rebuilt inside the timed region, nothing is carried over
        _lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(V), dp(tab)))
        _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V),
                                                  dp(dok)))
        ev[2].record()
        _lib.check(L.bsx_dev_commit_tally(ctx, st, dp(dv), C.c_uint32(nh), C.c_uint32(V), dp(dhh), dp(dok), dp(dres)))
        ev[3].record()
    once()
    torch.cuda.synchronize(dev)
    assert int(dok.sum().item()) == n, "stress: a signature failed to verify"
    reps = 3
    t_sha = t_ed = t_tally = 0.0
    for _ in range(reps):
        once()
        torch.cuda.synchronize(dev)
        t_sha += ev[0].elapsed_time(ev[1]); t_ed += ev[1].elapsed_time(ev[2]); t_tally += ev[2].elapsed_time(ev[3])
    t_sha, t_ed, t_tally = t_sha / reps, t_ed / reps, t_tally / reps
    tot = t_sha + t_ed + t_tally
    return {"workload": f"mode S: {nh} headers x {V} validators = {n} signatures (one header_range_{nh}, a commit per header)",
            "headers_per_s": nh / tot * 1e3, "ed25519_verifies_per_s": n / t_ed * 1e3,
            "sha512_challenge": {"ms": t_sha, "algorithmic_GBps": n * 237 / t_sha / 1e6, "frac_of_hbm_peak": n * 237 / t_sha / 1e6 / HBM_PEAK_GBS,
                                 "bytes_per_unit": 237},
            "ed25519_ms": t_ed, "ed25519_path": "fixed-key tables (built inside the timed region) + keyed verify",
            "tally_validator_hash_ms": t_tally}


def secondary_1024(args):
    """The metric's other production shape, header_range_1024 (32 map jobs x 32 headers, bin/header_range_1024.rs:6-17),
    through the same code path: this script again with --batch 32 in a fresh process (streams, hardware queues and the
    allocator start clean), same ranges per step, half the slots.  Reported beside the headline, N = 1 only."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--jobs", "32", "--batch", "32", "--validators", str(args.validators),
           "--ranges", str(args.ranges), "--engines", str(args.engines), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--no-stress"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        return {"error": out.stderr[-500:]}
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    return {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
            "steps": d["steps"], "roofline_frac": d["roofline"]["frac"]}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # test hooks (tests/test_gpu_engine.py runs the N > 1 code path with two ranks on ONE GPU): device override and a
    # gloo process group; the driver's multi-GPU runs use neither (one rank per GPU over RCCL)
    if os.environ.get("BSX_BENCH_DEVICE") is not None:
        local = int(os.environ["BSX_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BSX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import synth
    from blobstreamx_amd.engine import PipelinedEngines

    J, B, V, R = args.jobs, args.batch, args.validators, args.ranges
    t0 = time.perf_counter()
    w = synth.Workload(4, R * world, J, B, v=V)          # config #4 seed; identical on every rank
    t_gen = time.perf_counter() - t0
    E = args.engines
    eng = PipelinedEngines(J, B, V, R, n_engines=E, rank=rank, world=world, device=dev, with_witness=not args.no_witness)
    eng.upload_workload(w)

    # correctness gate before timing: statuses clean, public output = (target header hash, commitment) for every owned range
    eng.step()
    res = eng.download()
    own = slice(rank * R, (rank + 1) * R)
    assert res["header_status"] == 0 and res["assemble_status"] == 0, res
    assert not res["range_status"].any() and not res["skip_status"].any(), (res["range_status"], res["skip_status"])
    assert (res["output64"][:, :32] == w.hashes[own, w.n_blocks]).all(), "target header hash mismatch"
    gpu_out64 = res["output64"].copy()

    def barrier():
        eng.join()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.step()
    barrier()
    t_sub = t_exp = 0.0
    t0 = time.perf_counter()
    pending = []
    for i in range(args.steps):
        if i % args.event_every == 0:
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(E)]
            eng.step(time_kernels=True, events=evs)
            pending.append(evs)
        else:
            eng.step()
    barrier()
    elapsed = time.perf_counter() - t0
    for step_evs in pending:             # HIP events on the launch stream of each engine; per-launch averages
        for evs in step_evs:
            t_sub += evs[0].elapsed_time(evs[1])
            if not args.no_witness:
                t_exp += evs[2].elapsed_time(evs[3])
    t_sub /= len(pending) * E
    t_exp /= len(pending) * E
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    headers_per_step = world * R * J * B
    value = headers_per_step / (elapsed / args.steps)

    if rank == 0:
        e0 = eng.engines[0]
        ml = e0.ml
        n_jobs = e0.RT * e0.jc           # map jobs per launch (one engine = 1/E of the step)
        # algorithmic bytes: expansion reads the compact witness once and writes 8 B per element
        exp_bytes = n_jobs * (int(ml["n_bytes"]) + 4 * int(ml["n_words"]) + int(ml["n_bools"]) + 8 * int(ml["n_elements"]))
        slots = n_jobs * B
        sub_bytes = slots * (362 + 352 + 64 + 32 + 64)      # per slot: proofs read; paths+curr, tuple, leaf hash, 2 tree nodes written
        out = {
            "metric": "headers/sec witness-gen, header_range_2048 (SHA HBM GB/s vs roofline in `roofline`/`kernels`)",
            "value": value, "unit": "headers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"header_range_{J * B} ({J} map jobs x {B} headers), {V} validators, mode F (one target commit per range), "
                                   f"{R} ranges per GPU per step, Goldilocks witness {'off' if args.no_witness else 'materialised'}",
                       "ranges_per_gpu": R, "headers_per_step": headers_per_step, "pipelined_chunks": E,
                       "parallelism": f"{world} x ({J // world} of {J} map jobs per range), 1 all-gather of 128-B records" if world > 1 else "1 GPU",
                       "witness_bytes_per_step_per_gpu": int(E * n_jobs * 8 * int(ml["n_elements"])) if not args.no_witness else 0,
                       "input_generation_s": round(t_gen, 2),
                       "ed25519_path": e0.ed_path, "commit_beside": e0.commit_with,
                       # setup-time choice of the witness buffer among a few allocations (engine._place_witness)
                       "witness_placement_probe": e0.placement_probe},
        }
        if not args.no_witness:
            # the same kernel alone on an idle GPU (after the timed region): what the overlap with the other chunk's hashing costs it
            import ctypes as C
            from blobstreamx_amd import _lib
            iso = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            t_iso = 0.0
            for _ in range(5):
                iso[0].record()
                _lib.check(e0.L.bsx_dev_expand_witness(e0.ctx, st, _lib.p(e0._ml), C.c_uint32(n_jobs), _lib.dp(e0.compact), _lib.dp(e0.witness_map)))
                iso[1].record()
                torch.cuda.synchronize(dev)
                t_iso += iso[0].elapsed_time(iso[1]) / 5
            # HBM bytes per map job from the PMC passes in profiles/r1_pmc_hbm_traffic.csv (WRITE_SIZE 14,925,404 KB +
            # FETCH_SIZE 200,228 KB for a launch of 4096 jobs = one pipelined chunk; rocprofv3 units are KB; FETCH_SIZE not
            # doubled: the kernel reads its source with dword loads)
            pmc_bytes_per_job = (14925404 + 200228) * 1024 / 4096
            out["roofline"] = {"kernel": "k_expand_witness (map-job section)", "bound": "hbm", "achieved": exp_bytes / t_exp / 1e6,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": exp_bytes / t_exp / 1e6 / HBM_PEAK_GBS,
                               "traffic": pmc_bytes_per_job * n_jobs,
                               "avg_launch_ms": t_exp, "algorithmic_bytes_per_launch": exp_bytes,
                               "isolated": {"avg_launch_ms": t_iso, "achieved": exp_bytes / t_iso / 1e6, "frac": exp_bytes / t_iso / 1e6 / HBM_PEAK_GBS},
                               "note": "expanded (witness-emitting) byte count: 8 B written per Goldilocks element + the compact read; "
                                       "`achieved` is measured inside the timed region where the kernel co-runs with the other chunk's "
                                       "ALU-bound hashing; `isolated` is the same launch alone; `traffic` = PMC bytes (profiles/) scaled per map job"}
        out["kernels"] = [{"kernel": "k_prove_subchain", "avg_launch_ms": t_sub, "slots_per_launch": slots,
                           "compact_bytes_per_slot": 874, "achieved_GBps": sub_bytes / t_sub / 1e6,
                           "frac_of_hbm_peak": sub_bytes / t_sub / 1e6 / HBM_PEAK_GBS,
                           "sha256_compressions_per_s": slots * 23 / t_sub * 1e3,
                           "note": "compact bytes (362 B proofs in + 512 B digests/tuple out per slot); integer-ALU bound, see DESIGN.md"}]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, J, B, V, args.cpu_seconds, gpu_out64)
        if world == 1 and not args.no_stress:
            out["stress"] = stress(args, dev)
            out["header_range_1024"] = secondary_1024(args)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
