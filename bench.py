#!/usr/bin/env python3
"""bench.py — headers/sec of header_range witness generation on MI355X (BASELINE.json metric).

A "step" = one pass of the whole hot path (header hashing + hint assembly + prove_subchain + reduce + final
asserts + target-commit verification + Goldilocks witness expansion) over one batch of R synthetic header_range_2048
instances (32 map jobs x 64 headers, 100 validators, mode F = one commit per range, exactly what one reference proof
does) whose inputs are already resident in HBM.  The timed object is the C-ABI pipeline (`bsx_pipeline_step`,
csrc/pipeline.hip) — Python only calls it.  value = N * R * 2048 headers / step time, all ranks, max over ranks.

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): every rank computes its 32/N-job slice of all
ranges, one all-gather of 128-byte records per pipelined chunk, the owner finishes its ranges.  --scaling weak (default):
R ranges per GPU (N * R in total); --scaling strong: R ranges in total, i.e. BASELINE config #4 literally (each
header_range_2048 split into N sub-ranges of 2048 / N headers).  Every rank re-proves a sample of its owned ranges with
an un-sharded pipeline on its own GPU and compares outputs, statuses, commit results, map-job records and a sampled
witness, so a wrong collective cannot go unnoticed.  --mode S: a commit on EVERY header (BASELINE config #5), the commits
sharded with their headers, one all-gather of 128-byte folds.

Every `peak` of an ALU-bound roofline is measured in THIS process on THIS device (`bsx_calibrate`, ~50 ms at start-up):
no constants carried over from another box.

Objects on the JSON line beside the contract's keys (N = 1 unless noted):
  roofline            dominant kernel of the headline (witness expansion, HBM-write bound); also at N > 1
  calibration         the device ceilings measured at start-up
  kernels             the SHA kernels' compact-byte rates (never mixed with the expanded figure)
  cpu_baseline        the C oracle timed on this box's host cores on a bounded sample (rank 0; also at N > 1); also the
                      checker of the timed pipeline's outputs AND of the full Goldilocks witness of sampled ranges
                      (config.witness_checked_ranges).  oracle/ is imported only by the cpu_baseline* functions and the
                      `cpu_baseline` legs of stress / fused_commitment — always as the checker / CPU timing, never on the
                      measured GPU path
  compact_only        the same step without the Goldilocks expansion (fresh process): the ALU-bound rate of the SHA path
  stress              mode S: Ed25519 + SHA-512 bound, with its own `roofline` (bound "valu") and `cpu_baseline`
  latency             ONE range through the host tier (bsx_header_range, host pointers in, 64 B out)
  with_input_upload   the headline step with the headers streamed from pinned host memory every step (PCIe inclusive)
  fused_commitment    Poseidon Merkle caps of the witness straight from the compact bytes (no 64x image): the pipeline's
                      BSX_PIPE_CAPS mode (headers/s) and the kernel alone vs the materialised form
  header_range_1024   the metric's other production shape
"""
import argparse
import csv
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# Field operations of ONE fixed-key Ed25519 verification (ed25519.h ed25519_verify_keyed_core: affine tables, 22 radix-4096
# digits of h for the key, 16 radix-65536 digits of s for B): 38 mixed additions (3 + 4 mul each, the last one 3 + 3), no
# doubling + encoding.  With the batch-inversion scratch (k_ed25519_finish) the encoding costs 5 multiplications per
# signature plus one inversion (254 sq + 11 mul) per 8 / 16 / 32 signatures — counted at 16.
# Round 5: a resident validator set's tables may hold 16-bit digits (BSX_COMMITS_KEYTABLE_WIDE): 16 + 16 = 32 additions.
def fe_mul_per_verify(kt_bits=12):
    adds = (253 + kt_bits) // kt_bits + 16
    return adds * 7 - 1 + 5 + 11 / 16
FE_MUL_PER_VERIFY = fe_mul_per_verify(12)
FE_SQ_PER_VERIFY = 254 / 16
# Goldilocks multiplications of one Poseidon permutation that NO formulation can avoid: the x^7 S-boxes (4 multiplications
# each: x2, x3 = x2*x, x4 = x2*x2, x7 = x4*x3) of 8 full rounds x 12 lanes + 22 partial rounds x 1 lane.  The MDS layers are
# multiplications by small constants (shifts/adds here) and are NOT counted: an upper-bound style ceiling, never below truth.
GL_MUL_PER_PERMUTATION = 4 * (8 * 12 + 22)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-autotune", action="store_true", help="keep the pipeline's default stream assignment")
    ap.add_argument("--ranges", type=int, default=256, help="header_range instances per GPU per step (R); with --scaling strong: in total")
    ap.add_argument("--jobs", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--validators", type=int, default=100)
    ap.add_argument("--engines", type=int, default=2, help="chunks of the step pipelined on separate HIP streams inside the library (the "
                    "ALU-bound hashing of one chunk beside the HBM-bound expansion of the other)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--mode", choices=["F", "S"], default="F", help="F: one commit per range (a reference proof); S: a commit on every header")
    ap.add_argument("--wide-tables", choices=["auto", "0", "1"], default="auto", help="mode S: 16-bit digits in the key tables (BSX_COMMITS_KEYTABLE_WIDE; "
                    "auto = on)")
    ap.add_argument("--no-witness", action="store_true", help="skip the Goldilocks expansion (reported as such)")
    ap.add_argument("--caps", action="store_true", help="Poseidon Merkle caps of the map-job witnesses from the compact bytes (with --no-witness: instead of the expansion)")
    ap.add_argument("--alternate", type=int, default=1, help="K buffer sets inside the pipeline, step i on set i mod K (pipelining across steps; the compact-only leg uses 2)")
    ap.add_argument("--merkle-wgs", type=int, default=0, help="bsx_pipeline_config.tune_merkle_workgroups (experiments; 0 = automatic)")
    ap.add_argument("--subchain-form", type=int, default=0, help="bsx_pipeline_config.tune_subchain (experiments; 0 = automatic)")
    ap.add_argument("--no-commit", action="store_true", help="experiments: no target-commit verification (not a valid headline)")
    ap.add_argument("--no-units", action="store_true", help="A/B: do not materialise the COMMIT / SKIP units (BSX_PIPE_NO_UNITS: round 3's witness)")
    ap.add_argument("--only-leg", choices=["coalescing"], default=None, help="run ONE secondary leg in this (fresh) process and print its JSON: "
                    "`coalescing` = latency.concurrent + hint_concurrent (the parent runs it as a subprocess before it creates its own queues)")
    ap.add_argument("--long-steps", type=int, default=200, help="steps of the second, longer timed loop reported as `long_run` (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only: none of the secondary objects")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dry-run", action="store_true", help="launcher check only: spawn / join the ranks, one all-gather over the process group, "
                    "ONE JSON line with the world that really ran — no GPU work (CPU-only boxes: BSX_DIST_BACKEND=gloo)")
    ap.add_argument("--collective", choices=["rccl-c", "torch"], default="rccl-c", help="N > 1: the all-gather of the data path is ncclAllGather "
                    "called by the library (bsx_pipeline_set_rccl; default) or torch.distributed through a callback")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): re-run this very command under
    torch.distributed.run with N ranks on this node, one per GPU, and hand its exit code on — so that the multi-GPU bench cannot
    silently run as N = 1.  Returns only when nothing had to be launched."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    if not args.dry_run and os.environ.get("BSX_BENCH_DEVICE") is None:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: no launcher in the environment — starting {args.gpus} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=dict(os.environ, BSX_BENCH_SELF_LAUNCHED="1")).returncode)


def dry_run(args, rank, world):
    """--dry-run: the launcher contract without a GPU — every rank joins the process group, one all-gather of (rank, pid), rank 0
    prints what really ran."""
    import torch.distributed as dist
    ranks = [(rank, os.getpid())]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("BSX_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
        got = [None] * world
        dist.all_gather_object(got, (rank, os.getpid()))
        ranks = got
    assert world == args.gpus, f"--gpus {args.gpus} but {world} rank(s) are running"
    assert sorted(r for r, _ in ranks) == list(range(world)) and len({p for _, p in ranks}) == world
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [r for r, _ in ranks], "processes": len({p for _, p in ranks}),
                          "launched_by": "bench.py itself" if os.environ.get("BSX_BENCH_SELF_LAUNCHED") else "the caller / an external launcher"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def log(msg):
    """progress on stderr (stdout carries the ONE JSON line)"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU legs and what the box really grants: the GPU boxes show 256 logical CPUs but run the container
    under a cgroup CPU quota (cpu.max 1600000/100000 = 16 CPUs): 256 threads then thrash the quota (152 k Ed25519 verifies/s
    vs 278 k at 32 threads, tools/cpu_probe.py).  -> (threads to use, description)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        pass
    model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    if quota and quota < n:
        t = max(1, min(n, int(round(2 * quota))))
        return t, f"{t} threads of {model} (cgroup CPU quota {quota:g} CPUs of {n} logical)"
    return n, f"{n} threads of {model} (no CPU quota)"


# ---------------------------------------------------------------------------------------------------------------- ceilings
class Calibration(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("valu_add_u32_lane_ops_per_s", "valu_mad_u64_u32_lane_ops_per_s", "valu_alignbit_lane_ops_per_s",
                                          "sha256_compress_per_s", "sha512_compress_per_s", "fe25519_mul_per_s", "fe25519_sq_per_s",
                                          "goldilocks_mul_per_s", "hbm_store_bytes_per_s")] + [("compute_units", C.c_uint32), ("clock_mhz", C.c_uint32)]


def calibrate(dev):
    """bsx_calibrate on this device: the ALU / store ceilings every ALU-bound `roofline.peak` below is priced against."""
    from blobstreamx_amd import _lib
    c = Calibration()
    _lib.check(_lib.lib().bsx_calibrate(_lib.context(dev.index or 0), C.byref(c)))
    d = {k: getattr(c, k) for k, _ in Calibration._fields_}
    d["source"] = "bsx_calibrate in this process (library's own device functions alone at 8 waves/SIMD; csrc/calibrate.hip)"
    return d


def keyed_verify_peak(cal, kt_bits=12):
    """Ed25519 verifications/s if only the kernel's field multiplications and squarings cost time, at the measured rates."""
    return 1.0 / (fe_mul_per_verify(kt_bits) / cal["fe25519_mul_per_s"] + FE_SQ_PER_VERIFY / cal["fe25519_sq_per_s"])


def valu_insts(kernel_substr):
    """Wave-level VALU instructions per unit of work of a kernel, from the committed SQ counter summaries (profiles/
    valu_insts.json, written by tools/valu_insts.py from `rocprofv3 --pmc SQ_INSTS_VALU` passes).  None if not profiled."""
    path = os.path.join(ROOT, "profiles", "valu_insts.json")
    if not os.path.exists(path):
        return None
    for k, v in json.load(open(path)).get("kernels", {}).items():
        if kernel_substr in k:
            return v
    return None


def valu_issue(cal, kernel_substr, units, seconds):
    """valu_issue_frac = SQ_INSTS_VALU (wave instructions, scaled to this launch) / (time x the measured v_add_u32 wave-issue
    rate of the device).  Independent of any body micro-benchmark: how much of the VALU issue bandwidth the kernel used."""
    v = valu_insts(kernel_substr)
    if not v or seconds <= 0:
        return None
    insts = v["wave_valu_insts_per_unit"] * units
    rate = cal["valu_add_u32_lane_ops_per_s"] / 64.0
    return {"valu_issue_frac": insts / (seconds * rate), "wave_valu_insts_per_unit": v["wave_valu_insts_per_unit"], "unit": v["unit"],
            "wave_issue_rate_per_s": rate, "counter_source": v.get("source")}


# ---------------------------------------------------------------------------------------------------------------- checkers
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
            "sha_ni": bool(oracle.has_shani())}


# ---------------------------------------------------------------------------------------------------------------- mode S
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


# ---------------------------------------------------------------------------------------------------------------- other legs
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
    D = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "hostcheck", "libconcdrive.so"))
    D.cd_header_range_loop.restype = C.c_double
    D.cd_hint_burst.restype = C.c_int
    return D


def concurrent_leg(dev, J, B, V, ks=(1, 2, 4, 8, 16, 32, 64), seconds=0.5, window_us=0, n_lanes=0, max_requests=0, pinned=False, serial=True):
    """The reference's shape of use: ONE range per `prove` call, several calls in flight under a multi-thread runtime
    (circuits/header_range.rs:180-181, bin/header_range_2048.rs:6-17).  K native threads call the UNCHANGED bsx_header_range back to
    back, each on its own range (pageable host memory in, 64 B out), for `seconds`:
      coalesced   all threads share ONE context with bsx_enable_coalescing (round 5): calls arriving together run as one launch set
      serial      every thread has its OWN context and the calls run one by one per context (round 4's shape), K = 1 and 16
    headers/s, p50 / p99 per call, requests per launch set; every output checked against the chain's own target hash."""
    import synth
    from blobstreamx_amd import _lib
    from blobstreamx_amd import batcher as BT
    L = _lib.lib()
    D = _concdrive()
    kmax = max(ks)
    n_w = min(kmax, 32)                                          # distinct ranges (threads beyond take them again)
    w = synth.Workload(4, n_w, J, B, v=V)
    cid = np.frombuffer(b"celestia", np.uint8).copy()
    inp = np.stack([np.frombuffer(w.input48(k % n_w), np.uint8) for k in range(kmax)]).copy()
    hdrs = [np.ascontiguousarray(w.headers[k % n_w]) for k in range(kmax)]
    if pinned:                                                   # page-locked caller memory: uploaded from where it lies (no staging copy)
        keep = [torch.empty(h.nbytes, dtype=torch.uint8, pin_memory=True) for h in hdrs]
        for t_, h in zip(keep, hdrs):
            t_.numpy()[:] = h.view(np.uint8).reshape(-1)
        hdrs = [t_.numpy().view(hdrs[0].dtype) for t_ in keep]
    tv = [np.ascontiguousarray(w.validators[k % n_w]) for k in range(kmax)]
    rv = [np.ascontiguousarray(w.trusted[k % n_w]) for k in range(kmax)]
    fh = np.array([int(w.first_height[k % n_w]) for k in range(kmax)], np.uint64)
    nh = np.array([hdrs[k].size for k in range(kmax)], np.uint64)
    lt = np.array([int(w.latest[k % n_w]) for k in range(kmax)], np.uint64)
    PP = C.c_void_p * kmax
    p_h, p_tv, p_rv = PP(*[h.ctypes.data for h in hdrs]), PP(*[x.ctypes.data for x in tv]), PP(*[x.ctypes.data for x in rv])
    cap = 1 << 16

    def run(ctx_handles, shared, K):
        lat = np.zeros((K, cap), np.float32)
        counts, rcs, o64 = np.zeros(K, np.int32), np.zeros(K, np.int32), np.zeros((K, 64), np.uint8)
        CT = C.c_void_p * len(ctx_handles)
        dt = D.cd_header_range_loop(CT(*[h.value for h in ctx_handles]), C.c_int(shared), C.c_int(K), C.c_double(seconds), C.c_uint32(J), C.c_uint32(B),
                                    C.c_uint32(V), _lib.p(inp), p_h, _lib.p(fh), _lib.p(nh), _lib.p(lt), p_tv, p_rv, _lib.p(cid), C.c_uint32(8), _lib.p(lat),
                                    C.c_int(cap), _lib.p(counts), _lib.p(o64), _lib.p(rcs))
        assert dt > 0 and not rcs.any(), (dt, rcs)
        for k in range(K):
            assert o64[k, :32].tobytes() == w.hashes[k % n_w, w.n_blocks].tobytes()
        allv = np.sort(np.concatenate([lat[k, :min(cap, counts[k])] for k in range(K)]))
        n = int(counts.sum())
        return {"threads": K, "calls": n, "headers_per_s": n * J * B / dt, "calls_per_s": n / dt, "p50_ms": float(allv[len(allv) // 2]),
                "p99_ms": float(allv[min(len(allv) - 1, int(len(allv) * 0.99))]), "h2d_GBps_implied": n * (J * B + 1) * 512 / dt / 1e9}

    # coalesced: one shared context
    shared_ctx = C.c_void_p()
    _lib.check(L.bsx_init(C.c_int(dev.index or 0), C.byref(shared_ctx)))
    cfg = BT.make_config(J, B, V, window_us=window_us, n_lanes=n_lanes, max_requests=max_requests)
    _lib.check(L.bsx_enable_coalescing(shared_ctx, C.byref(cfg)))
    L.bsx_context_batcher.restype = C.c_void_p
    view = BT.Batcher(J, B, V, handle=C.c_void_p(L.bsx_context_batcher(shared_ctx)))
    run([shared_ctx], 1, min(8, kmax))                           # warm: lanes, key tables
    rows = []
    for K in ks:
        s0 = view.stats()["header_range"]
        row = run([shared_ctx], 1, K)
        s1 = view.stats()["header_range"]
        nb = max(1, s1["batches"] - s0["batches"])
        row["requests_per_launch_set"] = (s1["requests"] - s0["requests"]) / nb
        # the worker's time per launch set by phase (us): collecting, staging (+ enqueuing the header uploads), enqueuing the kernels, waiting
        # for the GPU, completing the tickets
        row["worker_us_per_set"] = {k: round((s1[k] * s1["batches"] - s0[k] * s0["batches"]) / nb, 1)
                                    for k in ("close_wait_us", "stage_wait_us", "enqueue_us", "gpu_wait_us", "complete_us")}
        rows.append(row)
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
    return {"workload": f"K native threads x bsx_header_range (one header_range_{J * B}, {V} validators per call, pageable host pointers in, 64 B out)",
            "coalesced_shared_context": rows, "serial_own_contexts": serial_rows, "headers_page_locked": bool(pinned),
            "pcie_note": f"every call uploads {(J * B + 1) * 512 / 1e6:.2f} MB of headers: 100 M headers/s = 51 GB/s of H2D, the PCIe Gen5 x16 practical "
                         "ceiling (with_input_upload measures ~46 GB/s on these boxes) — h2d_GBps_implied says how close a row is",
            "note": "the reference proves ONE range per call under a multi-thread runtime (header_range.rs:180-181): this is that shape.  Round 4 "
                    "(serial, own contexts): 16 callers = 1.4x one caller; coalesced: concurrent calls share launch sets"}


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


def subprocess_leg(args, extra, timeout=900):
    cmd = [sys.executable, os.path.abspath(__file__), "--jobs", str(args.jobs), "--batch", str(args.batch), "--validators", str(args.validators),
           "--ranges", str(args.ranges), "--engines", str(args.engines), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-legs"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:
        return None, out.stderr[-500:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]), None


def subprocess_legs(args):
    """The legs that run in processes of their own — BEFORE this process creates its first HIP queue: a parent that still holds its
    pipelines', contexts' and torch's streams (a hundred queues after the concurrency leg) makes the firmware time-slice the
    processes' queue sets, and the compact step measured beside it came out 12 % slow (372 M headers/s against 425-433 M alone)."""
    out = {}
    J, B = args.jobs, args.batch
    log("leg: compact_only (subprocess)")
    # one chunk per step: without an expansion to run beside there is nothing to pipeline against, and a chunk of 256
    # ranges quantises better (8196 header groups on 4096 wave slots) than two of 128
    # ... and two pipelines stepped in turn: step i + 1 starts while step i's chain of small kernels drains
    # 200 steps (0.26 s): a step's chain of dependent kernels spans two to three steps of the pipelined loop, so a 20-step
    # loop would spend a tenth of its time filling and draining (405 M headers/s at 20 steps, 423 M at 200 .. 6000)
    d, err = subprocess_leg(args, ["--no-witness", "--engines", "1", "--alternate", "2", "--steps", str(max(args.steps, 200))])
    out["compact_only"] = {"error": err} if d is None else {
        "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
        "prove_subchain_ms": d["kernels"][0]["avg_launch_ms"], "sha256_compressions_per_s_prove_subchain": d["kernels"][0]["sha256_compressions_per_s"],
        "frac_of_measured_alu_peak_prove_subchain": d["kernels"][0]["frac_of_measured_alu_peak"],
        **d["compact_step"],
        "reference_equivalent_compressions_per_s_whole_step": d["value"] * (41 + 23),
        "note": "no Goldilocks expansion, one chunk per step, two buffer sets stepped in turn inside the pipeline (one header hashing at a time): header hashing (41 compressions/header) + prove_subchain + commit check "
                "(Ed25519, SHA-512) on the side stream; fractions are of the SHA-256 ceiling measured in that process"}
    log("leg: coalescing = latency.concurrent + hint_concurrent (subprocess)")
    # the callers' threads and the batcher's lanes in a process of their own: beside the parent's pipelines, sixteen contexts and torch's
    # streams the same legs measured 20 % lower (K = 16: 37 M against 46 M headers/s; the 32 hints 0.44 against 0.33 ms)
    d, err = subprocess_leg(args, ["--only-leg", "coalescing"])
    out["coalescing"] = {"error": err} if d is None else d
    if not args.no_witness:
        # A/B on THIS box: the headline step with and without the COMMIT / SKIP units (two more expansion launches per chunk, the field
        # proofs and the unit stores of the commit chain on the side stream) — VERDICT r4 weak #8 asked what they cost the big expansion
        log("leg: units_ab (2 subprocesses)")
        ab = {}
        for key, extra in (("with_units", []), ("without_units", ["--no-units"])):
            d, err = subprocess_leg(args, ["--long-steps", "0"] + extra)
            ab[key] = {"error": err} if d is None else {
                "value": d["value"], "ms_per_step": d["ms_per_step"], "expand_map_avg_launch_ms": d["roofline"]["avg_launch_ms"],
                "frac_of_measured_store_ceiling": d["roofline"]["frac_of_measured_store_ceiling"],
                "stored_bytes_per_step": d["roofline_whole_step"]["stored_bytes_per_step"], "whole_step_store_GBps": d["roofline_whole_step"]["achieved"]}
        ab["note"] = ("same process shape, same box, back to back: `without_units` = BSX_PIPE_NO_UNITS (round 3's witness: map jobs + reduce nodes); the "
                      "difference in expand_map_avg_launch_ms is what the units' side-stream work costs the large launch")
        out["units_ab"] = ab
    if (J, B) == (32, 64):
        log("leg: header_range_1024 (subprocess)")
        a1024 = argparse.Namespace(**vars(args))
        a1024.batch = 32
        d, err = subprocess_leg(a1024, [])
        out["header_range_1024"] = {"error": err} if d is None else {
            "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
            "steps": d["steps"], "roofline_frac": d["roofline"]["frac"], "witness_checked_ranges": d["config"]["witness_checked_ranges"],
            # BASELINE config #3 ("header_range_1024 ... with rocprof HBM-GB/s counters"): the leg's own roofline object, `traffic` from the
            # committed PMC passes of THIS shape (profiles/r5_1024_pmc_hbm_traffic.csv)
            "roofline": d["roofline"]}
    return out


def pmc_traffic(n_units, match):
    """HBM bytes of one k_expand_witness launch of `n_units` units from the committed rocprofv3 PMC passes: the newest
    profiles/*pmc_hbm_traffic*.csv whose .meta.json (written by the profiling script) matches `match` — {"batch": B} for the map-job
    section of a header_range_{32 B}, {"layout": "commit", "v": V} for mode S's COMMIT units — scaled per unit.  (None, None) when no
    profile of that shape is committed."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_hbm_traffic*.meta.json")), reverse=True)      # r5 before r4 ...
    for meta in cands:
        m = json.load(open(meta))
        m.setdefault("layout", "map")
        want = dict({"layout": "map"}, **match)
        if any(m.get(k) != v for k, v in want.items()):
            continue
        path = meta[:-len(".meta.json")] + ".csv"
        if not os.path.exists(path):
            continue
        kb = {}
        for r in csv.DictReader(open(path)):
            if "k_expand_witness" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                kb[r["counter"]] = max(kb.get(r["counter"], 0), int(r["per_launch_max"]))
        if len(kb) == 2:
            # rocprofv3 units are KB; FETCH_SIZE not doubled: the kernel reads its source with dword loads (guide: HBM section)
            per_launch = int(m.get("jobs_per_launch") or m.get("units_per_launch"))
            return (kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024 / per_launch * n_units, os.path.basename(path)
    return None, None


def memory_partition_mode():
    for cmd in (["rocm-smi", "--showmemorypartition", "--showcomputepartition"], ["amd-smi", "partition"]):
        try:
            o = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            lines = [ln.strip() for ln in o.stdout.splitlines() if "artition" in ln and ":" in ln]
            if lines:
                return "; ".join(lines[:4])
        except Exception:
            pass
    return None


def sharded_self_check(eng, w, J, B, V, R, rank, dev, res):
    """N > 1 must prove itself: re-prove a sample of the owned ranges with an UN-SHARDED pipeline (world = 1, all map jobs,
    no collective) on this rank's own GPU and compare public output, final status, commit result, this rank's map-job
    records and — for one range — this rank's slice of the Goldilocks witness."""
    from blobstreamx_amd import engine as E
    from blobstreamx_amd import types as T
    ks = sorted({0, R // 2, R - 1})
    solo = E.HeaderRangeEngine(J, B, V, len(ks), device=dev, with_witness=eng.with_witness)
    solo.upload_workload(w, np.array([rank * R + k for k in ks]))
    solo.step()
    sres = solo.download()
    for i, k in enumerate(ks):
        assert sres["output64"][i].tobytes() == res["output64"][k].tobytes(), f"rank {rank}: sharded output of owned range {k} differs from the un-sharded pipeline"
        assert sres["range_status"][i] == res["range_status"][k] and sres["skip_status"][i] == res["skip_status"][k]
        a, b = np.array(sres["commit"][i]).copy(), np.array(res["commit"][k]).copy()
        a["_pad"] = 0; b["_pad"] = 0
        assert a.tobytes() == b.tobytes()
        mine = res["records"][rank * R + k].copy()
        ref = sres["records"][i][eng.jf:eng.jf + eng.jc].copy()
        mine["_pad"] = 0; ref["_pad"] = 0
        assert mine.tobytes() == ref.tobytes(), f"rank {rank}: map-job records of owned range {k} differ from the un-sharded pipeline"
    checked_w = 0
    if eng.with_witness:
        nel = int(T.map_layout(B)["n_elements"])
        sw = solo.witness_map[eng.jf * nel:(eng.jf + eng.jc) * nel]          # un-sharded range ks[0], jobs of this rank's slice
        i0 = eng.rank * eng.Rc                                               # owned range 0 = chunk 0, position rank*Rc
        mw = eng.buffer(0, E.BUF_WITNESS_MAP, i64=True)[i0 * eng.jc * nel:(i0 + 1) * eng.jc * nel]
        assert torch.equal(sw, mw), f"rank {rank}: sharded map-job witness of owned range 0 differs from the un-sharded pipeline"
        checked_w = 1
    del solo
    return {"ranges": len(ks), "fields": "output64, statuses, commit result, map-job records", "witness_ranges": checked_w}


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    self_launch(args)                                   # --gpus N without a launcher: spawns the N ranks and exits with their code
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, rank, world)
    # the line's n_gpus is the world that RUNS: a launcher's WORLD_SIZE wins over a stale --gpus, and N ranks need N GPUs
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
        args.gpus = world
    if os.environ.get("BSX_BENCH_DEVICE") is None:
        assert torch.cuda.device_count() >= world, f"{world} ranks on this node need {world} GPUs, {torch.cuda.device_count()} visible"
    # test hooks (tests/test_gpu_engine.py runs the N > 1 code path with two ranks on ONE GPU): device override and a
    # gloo process group; the driver's multi-GPU runs use neither (one rank per GPU over RCCL)
    if os.environ.get("BSX_BENCH_DEVICE") is not None:
        local = int(os.environ["BSX_BENCH_DEVICE"])
    pre_legs = subprocess_legs(args) if (world == 1 and args.mode == "F" and not args.no_legs) else {}
    from blobstreamx_amd import _lib
    _lib.lib()                     # loads libbsx.so and calls bsx_prepare_process() before the first HIP call (16 hardware queues)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BSX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import synth
    from blobstreamx_amd import engine as E
    from blobstreamx_amd import types as T

    if args.only_leg == "coalescing":
        J, B, V = args.jobs, args.batch, args.validators
        print(json.dumps({"concurrent": concurrent_leg(dev, J, B, V), "hint_concurrent": hint_concurrent_leg(dev, J, B, V)}))
        return
    cal = calibrate(dev)
    J, B, V = args.jobs, args.batch, args.validators
    if args.mode == "S":
        # mode S as the primary object: commits sharded with their headers across the ranks
        out = stress(args, dev, V, args.cpu_seconds, cal, rank=rank, world=world)
        if rank == 0:
            print(json.dumps({"metric": "headers/sec, mode S (a commit on every header)", "value": out["headers_per_s"], "unit": "headers/s",
                              "n_gpus": world, "steps": 5, "warmup": 1, "ms_per_step": out["ms"], "higher_is_better": True, "scaling": "strong",
                              "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                              "config": {"workload": out["workload"], "nccl_ranks": world, "dist_backend": backend,
                                         "parallelism": f"{world} x {args.jobs * args.batch // world} commits, 1 all-gather of 128-byte folds per step"},
                              "roofline": out["roofline"], "cpu_baseline": out.get("cpu_baseline"), "calibration": cal, "stress": out}))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    strong = args.scaling == "strong" and world > 1
    R = args.ranges // world if strong else args.ranges          # ranges OWNED per rank
    assert R >= args.engines and R % args.engines == 0, "ranges per rank must be a multiple of the pipelined chunks"
    t0 = time.perf_counter()
    w = synth.Workload(4, R * world, J, B, v=V)          # config #4 seed; identical on every rank
    t_gen = time.perf_counter() - t0
    Ech = args.engines
    kw = dict(n_engines=Ech, rank=rank, world=world, device=dev, with_witness=not args.no_witness, with_caps=args.caps, merkle_workgroups=args.merkle_wgs,
              with_commit=not args.no_commit, subchain_form=args.subchain_form, units=not args.no_units)
    eng = E.AlternatingPipelines(args.alternate, J, B, V, R, **kw) if args.alternate > 1 else E.PipelinedEngines(J, B, V, R, **kw)
    eng.upload_workload(w)
    p0 = eng
    collective = None
    if world > 1:
        # The one collective of the path.  Default: ncclAllGather called by the LIBRARY on its exchange stream, on a communicator made
        # through the C ABI (rank 0's unique id goes over the torch.distributed group: control plane only) — no Python between the
        # local fold and the top fold.  Every rank must take the same branch: the availability probe is agreed on first, and
        # bsx_pipeline_check_allgather proves the collective end to end before the first step.
        import torch.distributed as dist
        collective = "torch.distributed all_gather_into_tensor through the bsx_pipeline_set_allgather callback"
        if args.collective == "rccl-c" and backend == "nccl":
            probe = np.zeros(128, np.uint8)
            ok = torch.tensor([1 if _lib.lib().bsx_rccl_get_unique_id(_lib.p(probe)) == 0 else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                # every step of the set-up is agreed on by all ranks before the next (a rank that fails alone would leave the others in a
                # collective): communicator, then the library's own all-gather proven end to end; any failure anywhere -> the callback
                def agreed(flag):
                    t = torch.tensor([1 if flag else 0], device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return int(t.item()) == 1
                comm, err = None, None
                try:
                    comm = E.c_rccl_comm(eng.ctx, rank, world)
                except Exception as e:                                   # noqa: BLE001 (reported below, then the fallback)
                    err = e
                if agreed(comm is not None):
                    try:
                        eng.set_rccl(comm)
                    except Exception as e:                               # noqa: BLE001
                        err = e
                    if agreed(err is None):
                        collective = "ncclAllGather called by libbsx (bsx_pipeline_set_rccl), communicator from bsx_rccl_comm_init_rank"
                if not collective.startswith("ncclAllGather"):
                    if rank == 0:
                        print(f"bench.py: RCCL from the C tier not usable here ({err}); using the torch.distributed callback", file=sys.stderr)
                    eng.set_allgather(E.torch_allgather(eng.dev, world))

    # correctness gate before timing: statuses clean, public output = (target header hash, commitment) for every owned range
    eng.step()
    res = eng.download()
    own = slice(rank * R, (rank + 1) * R)
    assert res["header_status"] == 0 and res["assemble_status"] == 0, res
    assert not res["range_status"].any() and (args.no_commit or not res["skip_status"].any()), (res["range_status"], res.get("skip_status"))
    assert args.no_commit or (res["output64"][:, :32] == w.hashes[own, w.n_blocks]).all(), "target header hash mismatch"
    gpu_out64 = res["output64"].copy()
    self_check = sharded_self_check(p0, w, J, B, V, R, rank, dev, res) if world > 1 else None

    def barrier():
        eng.join()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # one-time, before the warm-up: which hardware queues the chunks' streams overlap best on (bsx_pipeline_autotune; real steps)
    tune = eng.autotune(0) if not args.no_autotune else None
    for _ in range(args.warmup):
        eng.step()
    barrier()
    eng.set_timing(True)             # HIP events on the launch streams, inside the library
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.step()
    barrier()
    elapsed = time.perf_counter() - t0
    tm = eng.timing()
    eng.set_timing(False)
    t_sub, t_exp = tm["prove_subchain_ms"], tm["expand_map_ms"]
    multi_gpu = None
    if world > 1:
        # what the first hardware scaling run needs to be diagnosable: every rank's own step time and the all-gather as the library
        # timed it on its exchange stream (HIP events), gathered to rank 0; `value` uses the MAX over ranks
        mine = torch.tensor([elapsed / args.steps * 1e3, tm["allgather_us"]["avg"], tm["allgather_us"]["min"], tm["allgather_us"]["median"],
                             tm["allgather_us"]["max"], float(tm["exchanges"])], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        rows = [[float(x) for x in t.tolist()] for t in allr]
        multi_gpu = {"per_rank_ms_per_step": [r[0] for r in rows],
                     "allgather_us_per_chunk": {"avg": [r[1] for r in rows], "min": [r[2] for r in rows], "median": [r[3] for r in rows],
                                                "max": [r[4] for r in rows]},
                     "allgathers_timed_per_rank": [int(r[5]) for r in rows],
                     "note": "all-gather of one 128-byte record per (range, rank) per chunk, timed with HIP events on the library's exchange stream from "
                             "'this rank's folded records are ready' to 'every rank's have arrived' (includes waiting for the slowest rank); it is "
                             "issued behind the local fold and joined behind the chunk's expansion, so it is hidden unless it outlasts the expansion"}
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
        assert torch.distributed.get_world_size() == world == args.gpus, (torch.distributed.get_world_size(), world, args.gpus)
    ms_per_step = elapsed / args.steps * 1e3
    headers_per_step = world * R * J * B
    value = headers_per_step / (elapsed / args.steps)
    # a second, longer loop (20 steps are 0.12 s — SURVEY §8(d)'s 100 ms with no margin): same steps, `long_run`
    long_run = None
    if args.long_steps > 0 and world == 1:
        torch.cuda.synchronize(dev)
        t0l = time.perf_counter()
        for i in range(args.long_steps):
            eng.step()
        eng.join()
        el = time.perf_counter() - t0l
        long_run = {"steps": args.long_steps, "seconds": el, "ms_per_step": el / args.long_steps * 1e3, "value": headers_per_step / (el / args.long_steps),
                    "unit": "headers/s"}
    # the witness the TIMED loop left in HBM, against the oracle (every rank checks its own buffers)
    n_checked = cpu_baseline_witness_check(p0, w, J, B) if not args.no_witness else 0
    res2 = eng.download()
    assert (res2["output64"] == gpu_out64).all() and not res2["range_status"].any() and (args.no_commit or not res2["skip_status"].any()), "outputs changed during the timed loop"

    if rank == 0:
        ml = p0.ml
        n_jobs = p0.RT * p0.jc           # map jobs per launch (one chunk = 1/E of the step)
        # algorithmic bytes: expansion reads the compact witness once and writes 8 B per element
        exp_bytes = n_jobs * (int(ml["n_bytes"]) + 4 * int(ml["n_words"]) + int(ml["n_bools"]) + 8 * int(ml["n_elements"]))
        slots = n_jobs * B
        sub_bytes = slots * (362 + 352 + 64 + 32 + 64)      # per slot: proofs read; paths+curr, tuple, leaf hash, 2 tree nodes written
        # every Goldilocks element the step stores (8 B each), by section: map jobs, reduce nodes (local + top levels), and — with the
        # commit check on — the COMMIT + SKIP unit of every owned range (the units were missing from this figure)
        rl_, cl_, sl_ = T.reduce_layout(), T.commit_layout(V), T.skip_layout(V)
        units_on = bool(getattr(p0, "units", False))
        step_sections = {"map_jobs": Ech * n_jobs * 8 * int(ml["n_elements"]),
                         "reduce_nodes": Ech * (p0.RT * max(p0.jc - 1, 0) + p0.Rc * (world - 1)) * 8 * int(rl_["n_elements"]),
                         "commit_units": (R * 8 * int(cl_["n_elements"])) if units_on else 0,
                         "skip_units": (R * 8 * int(sl_["n_elements"])) if units_on else 0}
        step_store_bytes = sum(step_sections.values())
        out = {
            "metric": f"headers/sec witness-gen, header_range_{J * B} (SHA HBM GB/s vs roofline in `roofline`/`kernels`)",
            "value": value, "unit": "headers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "long_run": long_run,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"header_range_{J * B} ({J} map jobs x {B} headers), {V} validators, mode F (one target commit per range), "
                                   f"{R} ranges per GPU per step, Goldilocks witness {'off' if args.no_witness else 'materialised'}"
                                   + (", Poseidon caps from the compact bytes" if args.caps else ""),
                       "timed_entry": "bsx_pipeline_step (C ABI, csrc/pipeline.hip)",
                       "ranges_per_gpu": R, "headers_per_step": headers_per_step, "pipelined_chunks": Ech, "buffer_sets": args.alternate,
                       "parallelism": (f"{world} x ({J // world} of {J} map jobs = {J * B // world} headers of every range), 1 all-gather of 128-B "
                                       f"records per chunk; {'strong: ' + str(R * world) + ' ranges in total' if strong else 'weak: ' + str(R) + ' ranges per GPU'}")
                       if world > 1 else "1 GPU",
                       "nccl_ranks": torch.distributed.get_world_size() if world > 1 else 1, "dist_backend": backend, "collective": collective,
                       "multi_gpu": multi_gpu,
                       "sharded_vs_unsharded_self_check_per_rank": self_check,
                       "witness_checked_ranges": n_checked,
                       "witness_bytes_per_step_per_gpu": int(step_store_bytes) if not args.no_witness else 0,
                       "witness_bytes_per_step_per_gpu_by_section": step_sections if not args.no_witness else None,
                       "input_generation_s": round(t_gen, 2), "stream_autotune": tune,
                       "ed25519_path": p0.ed_path, "commit_beside": p0.commit_with, "memory_partition": memory_partition_mode()},
            "calibration": cal,
        }
        if not args.no_witness:
            # the same kernel alone on an idle GPU (after the timed region): what the overlap with the other chunk's hashing costs it
            iso = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            compact, wmap = p0.buffer(0, E.BUF_COMPACT), p0.buffer(0, E.BUF_WITNESS_MAP, i64=True)
            t_iso = 0.0
            for _ in range(5):
                iso[0].record()
                _lib.check(p0.L.bsx_dev_expand_witness(p0.ctx, st, _lib.p(p0._ml), C.c_uint32(n_jobs), _lib.dp(compact), _lib.dp(wmap)))
                iso[1].record()
                torch.cuda.synchronize(dev)
                t_iso += iso[0].elapsed_time(iso[1]) / 5
            traffic, traffic_src = pmc_traffic(n_jobs, {"batch": B})
            # SURVEY §8(d)'s LITERAL per-slot figure (362 B read + (362 + 384) x 64 B written = 48,106 B per header slot) beside the
            # layout's own count (every variable bsx_witness_manifest lists: 57,124 B per slot at B = 64) — VERDICT r3 weak #8
            survey_bytes = slots * 48106
            out["roofline"] = {"kernel": "k_expand_witness (map-job section)", "bound": "hbm", "achieved": exp_bytes / t_exp / 1e6,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": exp_bytes / t_exp / 1e6 / HBM_PEAK_GBS,
                               "byte_accounting": {"layout_bytes_per_slot": exp_bytes / slots, "frac_layout_bytes": exp_bytes / t_exp / 1e6 / HBM_PEAK_GBS,
                                                   "survey_8d_bytes_per_slot": 48106, "achieved_survey_8d": survey_bytes / t_exp / 1e6,
                                                   "frac_survey_8d": survey_bytes / t_exp / 1e6 / HBM_PEAK_GBS,
                                                   "note": "`frac` prices the bytes the kernel really moves (the layout's variables, = PMC traffic 1.00x); "
                                                           "SURVEY §8(d)'s literal figure counts 12 digests per slot where the layout holds the 10 path digests, "
                                                           "curr_header, the tuple, leaf hash, two tree nodes, words and bools"},
                               "traffic": traffic, "traffic_source": traffic_src,
                               "avg_launch_ms": t_exp, "algorithmic_bytes_per_launch": exp_bytes, "launches_timed": tm["launches"],
                               "measured_store_ceiling_GBps": cal["hbm_store_bytes_per_s"] / 1e9,
                               "frac_of_measured_store_ceiling": min(1.0, exp_bytes / t_exp * 1e3 / cal["hbm_store_bytes_per_s"]),
                               "isolated": {"avg_launch_ms": t_iso, "achieved": exp_bytes / t_iso / 1e6, "frac": exp_bytes / t_iso / 1e6 / HBM_PEAK_GBS},
                               "note": "expanded (witness-emitting) byte count: 8 B written per Goldilocks element + the compact read; "
                                       "`achieved` is measured with HIP events on the launch stream inside the timed region where the kernel "
                                       "co-runs with the other chunk's ALU-bound hashing; `isolated` is the same launch alone; `traffic` = PMC "
                                       "bytes per launch read from the committed profile at run time (null when no profile of this shape exists)"}
            # the same bound priced over the WHOLE step: every byte stored (map jobs + reduce nodes + COMMIT / SKIP units) over the step's
            # wall time — what `headers/s` does not credit (the units are 1.2 GB of the 30.7 GB per step)
            out["roofline_whole_step"] = {"kernel": "every k_expand_witness launch of a step (map jobs, reduce nodes, COMMIT + SKIP units)", "bound": "hbm",
                                          "achieved": step_store_bytes / ms_per_step / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": step_store_bytes / ms_per_step / 1e6 / HBM_PEAK_GBS, "stored_bytes_per_step": int(step_store_bytes),
                                          "sections": step_sections, "ms_per_step": ms_per_step,
                                          "frac_of_measured_store_ceiling": min(1.0, step_store_bytes / ms_per_step * 1e3 / cal["hbm_store_bytes_per_s"]),
                                          "traffic": None, "note": "stores only (the compact reads are 1.6 % of the traffic); wall time of the step, so the "
                                                                   "hashing / commit phases that run beside the expansions are inside it"}
        else:
            out["roofline"] = None
        fused = bool(p0.fused_hint)
        exec_per_slot = (2 + 2 * (B - 1) / B) if fused else (21 + 2 * (B - 1) / B)     # tuple leaf + tree (+ both proof paths)
        comp_s = slots * exec_per_slot / t_sub * 1e3
        one_launch = fused and not (p0.with_witness and p0.E > 1)
        out["kernels"] = [{"kernel": "prove_subchain (k_batch_finish<fused>: tuple leaf hashes + every tree level + predicates in one launch)" if one_launch
                           else "prove_subchain (k_slot_hashes + k_tree_level x n + k_batch_finish)", "avg_launch_ms": t_sub, "slots_per_launch": slots,
                           "compact_bytes_per_slot": 874, "achieved_GBps": sub_bytes / t_sub / 1e6,
                           "frac_of_hbm_peak": sub_bytes / t_sub / 1e6 / HBM_PEAK_GBS,
                           "fused_hint": fused, "sha256_compressions_executed_per_slot": exec_per_slot,
                           "sha256_compressions_per_s": comp_s, "frac_of_measured_alu_peak": min(1.0, comp_s / cal["sha256_compress_per_s"]),
                           "reference_equivalent_compressions_per_s": slots * 23 / t_sub * 1e3,
                           "note": "compact bytes (362 B proofs in + 512 B digests/tuple out per slot); integer-ALU bound; peak = the SHA-256 compression "
                                   "rate measured in this run.  With the fused hint the 19 path compressions per slot the reference's circuit performs "
                                   "(builder.rs:189-199) are NOT executed: their digests are nodes of the header trees k_header_merkle hashed (41 "
                                   "compressions/header) and are copied, so `reference_equivalent` counts the reference's 23/slot over the same time; "
                                   "in-region = beside the other chunk's expansion"}]
        if args.no_witness:
            per_header = 41 + exec_per_slot
            out["compact_step"] = {"sha256_compressions_executed_per_header": per_header, "sha256_compressions_per_s_whole_step": value * per_header,
                                   "frac_of_measured_alu_peak_whole_step": min(1.0, value * per_header / cal["sha256_compress_per_s"]),
                                   "measured_sha256_ceiling_per_s": cal["sha256_compress_per_s"]}
        # rank 0 keeps the CPU baseline at every N (north_star: GPU throughput next to the CPU path, core count stated);
        # the other legs are N = 1 only
        log("headline timed; cpu_baseline")
        if not args.no_legs and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, J, B, V, args.cpu_seconds, gpu_out64, min(R, 256), first=rank * R)
        legs = world == 1 and not args.no_legs
        if legs and not args.no_witness:
            log("leg: with_input_upload")
            out["with_input_upload"] = upload_leg(p0, args, max(5, args.steps // 2))
        del eng, p0
        torch.cuda.empty_cache()
        if legs:
            log("leg: latency")
            out["latency"] = latency_leg(dev, J, B, V)
            co = pre_legs.pop("coalescing", None) or {}
            out["latency"]["concurrent"] = co.get("concurrent", co)
            out["hint_concurrent"] = co.get("hint_concurrent", co)
            log("leg: keyset_churn")
            out["keyset_churn"] = keyset_churn_leg(dev, J, B, V)
            log("leg: range_sweep")
            out["range_sweep"] = {"compact": range_sweep_leg(dev, J, B, V), "witness": range_sweep_leg(dev, J, B, V, rs=(1, 4, 16, 64), witness=True)}
            log("leg: fused_commitment")
            out["fused_commitment"] = commitment_leg(dev, J, B, V, cal)
            if not args.no_stress:
                log("leg: stress v100")
                sv100 = stress(args, dev, 100, 6.0, cal)
                log("leg: stress v512")
                out["stress"] = {"v100": sv100, "v512": stress(args, dev, 512, 6.0, cal)}
            out.update(pre_legs)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
