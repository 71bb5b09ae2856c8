"""Device-resident batched header_range pipeline (the throughput path bench.py times).

R independent header_range instances per pass, inputs already in HBM, every kernel enqueued through the device tier of
the C ABI (include/bsx.h, bsx_dev_*) on a main HIP stream plus a side stream for the commit check.  PyTorch only owns
the device buffers, the streams/events and — across GPUs — the one collective; no torch kernel runs on the data path.
PipelinedEngines splits a pass into two chunks whose hashing and expansion phases alternate.

Pass over R ranges of J map jobs x B headers (reference shapes 32x32 / 32x64, bin/header_range_{1024,2048}.rs:6-17):

  1 header_merkle     tendermint Header::hash + inclusion proofs for every supplied header   (input.rs:175-195,250-261)
  2 fill_end_hash     ctx.end_header_hash := target header hash (output of builder.skip)     (header_range.rs:42-55)
  3 commit            SHA-512 challenge -> Ed25519 -> tallies/validator hashes -> skip check  (header_range.rs:42-48)
  4 assemble_inputs   the hint of every map job                                               (data_commitment.rs:22-44)
  5 prove_subchain    map stage                                                               (builder.rs:150-271,305-336)
  6 reduce            local fold of this device's jobs, [all-gather across GPUs], top fold    (builder.rs:337-395)
  7 finalize          range check, final asserts, 64-byte public output                       (builder.rs:292-297,400-406)
  8 expand_witness    compact witness -> Goldilocks elements (map jobs + reduce nodes)

Multi-GPU: rank g owns map jobs [g*J/N, (g+1)*J/N) of EVERY range of the global batch (N*R ranges, so the
per-GPU slot count is the same as at N = 1: weak scaling), folds them locally, and ONE all-gather of a 128-byte
record per (range, rank) replaces the reference's map->reduce hand-off; the owner of a range (range index // R) does
the last log2(N) reduce levels, the final assertions and that range's commit verification.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import types as T


def _u8(n, dev):
    return torch.zeros(max(int(n), 16), dtype=torch.uint8, device=dev)


def job_slice(nb_map_jobs, rank, world):
    """Map jobs [first, first+count) of every range that `rank` owns (aligned power-of-two slices so the
    local fold + top fold is the same binary tree plonky2x mapreduce builds, circuits/builder.rs:301-302)."""
    assert nb_map_jobs % world == 0, "world size must divide NB_MAP_JOBS"
    count = nb_map_jobs // world
    assert count & (count - 1) == 0, "each rank needs a power-of-two slice of the map jobs"
    return rank * count, count


def all_gather_records(partial, world, n_ranges_total, out_gathered=None, async_op=False):
    """THE collective of the multi-GPU path: all-gather one 128-byte MapReduceSubchainVariable record per
    (range, rank) -> uint8 [world][n_ranges_total][128].  partial: this rank's locally folded record of every range.
    Works on CUDA tensors over RCCL ("nccl") and on CPU tensors over gloo (tests).
    async_op: return (gathered, work) without waiting; work.wait() orders the caller's stream behind the collective."""
    import torch.distributed as dist
    RT = n_ranges_total
    flat = partial[:RT * 128].contiguous()
    gathered = out_gathered[:world * RT * 128] if out_gathered is not None else torch.empty(world * RT * 128, dtype=torch.uint8, device=flat.device)
    work = None
    if flat.is_cuda and dist.get_backend() == "gloo":
        # test-only route (two ranks sharing one GPU, tests/test_gpu_engine.py): gloo moves host memory
        g_cpu = torch.empty(world * RT * 128, dtype=torch.uint8)
        dist.all_gather_into_tensor(g_cpu, flat.cpu())
        gathered.copy_(g_cpu)
    else:
        work = dist.all_gather_into_tensor(gathered, flat, async_op=async_op)
    return (gathered, work) if async_op else gathered


def gather_partials(partial, rank, world, n_ranges_local, out_gathered=None, out_top=None):
    """all_gather_records + the owned ranges laid out as [range][rank] (host-side consumers and the CPU tests; the
    device path folds the gathered layout in place with bsx_dev_reduce_strided)."""
    RT = world * n_ranges_local
    gathered = all_gather_records(partial, world, RT, out_gathered)
    g = gathered.view(world, RT, 128)
    own = g[:, rank * n_ranges_local:(rank + 1) * n_ranges_local, :]            # [rank, owned range, 128]
    top = out_top[:n_ranges_local * world * 128] if out_top is not None else torch.empty(n_ranges_local * world * 128, dtype=torch.uint8, device=gathered.device)
    top.view(n_ranges_local, world, 128).copy_(own.transpose(0, 1))
    return top


class HeaderRangeEngine:
    def __init__(self, nb_map_jobs, batch_size, v_max, n_ranges_local, rank=0, world=1, device=None, with_witness=True,
                 with_commit=True, chain_id=b"celestia"):
        self.J, self.B, self.V = nb_map_jobs, batch_size, v_max
        self.chain_id = np.frombuffer(bytes(chain_id), np.uint8).copy()       # C::CHAIN_ID_BYTES (header_range.rs:42-43)
        self.rank, self.world = rank, world
        self.R = n_ranges_local                     # ranges owned by this rank (commit + final reduce)
        self.RT = n_ranges_local * world            # ranges whose job slice this rank computes
        self.jf, self.jc = job_slice(nb_map_jobs, rank, world)   # first job / jobs per range on this rank
        self.with_witness, self.with_commit = with_witness, with_commit
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.ctx = _lib.context(self.dev.index if self.dev.index is not None else 0)
        self.L = _lib.lib()
        self.ml, self.rl = T.map_layout(batch_size), T.reduce_layout()
        self._ml = np.array(self.ml).reshape(1)
        self._rl = np.array(self.rl).reshape(1)
        d = self.dev
        B, jc, RT, R, V = batch_size, self.jc, self.RT, self.R, v_max
        self.hpr = jc * B + 1                       # headers this rank holds per range: its slice + the next one
        self.hfr = self.jf * B                      # height offset of the first of them (header_first_rel)
        # one header block per pass: this rank's slice of every range, then (owned ranges) the trusted and the target
        # header of the commit check as a 2-header block per range — hashed by ONE k_header_merkle launch
        nh_main, nh_skip = RT * self.hpr, (R * 2 if with_commit else 0)
        self.nh_all = nh_main + nh_skip
        self.headers_all = _u8(self.nh_all * 512, d)
        self.hashes_all = _u8(self.nh_all * 32, d)
        self.headers = self.headers_all[:nh_main * 512]
        self.hashes = self.hashes_all[:nh_main * 32]
        self.skip_headers = self.headers_all[nh_main * 512:] if nh_skip else _u8(16, d)
        self.skip_hashes = self.hashes_all[nh_main * 32:] if nh_skip else _u8(16, d)
        self.dh_aunts = _u8(self.nh_all * 128, d)
        self.lb_aunts = _u8(self.nh_all * 128, d)
        # "fused hint" (default): k_header_merkle also hands over the 7 path digests per header, the hint copies them into
        # the slots and prove_subchain does not re-derive them (19 of its 21 compressions per slot); BSX_FUSED_HINT=0 keeps
        # the proofs-only hand-over
        self.fused_hint = os.environ.get("BSX_FUSED_HINT", "1") != "0"
        self.paths = _u8(self.nh_all * 224, d) if self.fused_hint else None
        # bsx.h: BSX_SUBCHAIN_PATHS_FROM_HINT (1); PipelinedEngines adds BSX_SUBCHAIN_SEPARATE_LAUNCHES (2) beside an expansion
        self.subchain_flags = 1 if self.fused_hint else 0
        self.expand_done = None                    # set by step_final(expand_stream=...)
        self.ranges = _u8(RT * 80, d)
        self.latest = _u8(RT * 8, d)
        self.status = torch.zeros(8, dtype=torch.int32, device=d)       # [0] header, [1] assemble
        self.compact = _u8(RT * jc * int(self.ml["compact_stride"]), d)
        self.records = _u8(RT * jc * 128, d)
        self.partial = _u8(RT * 128, d)              # local fold: one record per range
        n_local_nodes = RT * max(jc - 1, 0)
        self.red_compact_local = _u8(n_local_nodes * int(self.rl["compact_stride"]), d)
        self.gathered = _u8(world * RT * 128, d)     # all-gather output [rank][range]
        self.red_compact_top = _u8(R * max(world - 1, 0) * int(self.rl["compact_stride"]), d)
        self.results = _u8(R * 128, d)
        self.output64 = _u8(R * 64, d)
        self.range_status = torch.zeros(max(R, 1), dtype=torch.int32, device=d)
        # commit (owned ranges): the trusted header and the target header as a 2-header block per range
        self.skip_ranges = _u8(R * 80, d)
        self.target_idx = torch.ones(max(R, 1), dtype=torch.int32, device=d)
        self.validators = _u8(R * V * 256, d)
        self.trusted = _u8(R * V * 256, d)
        self.h = _u8(R * V * 32, d)
        self.ok = _u8(R * V, d)
        self.commit_res = _u8(R * 96, d)
        self.trusted_res = _u8(R * 96, d)
        self.skip_status = torch.zeros(max(R, 1), dtype=torch.int32, device=d)
        # Commit-check inputs that the main stream produces are double-buffered by pass parity, so that the check of
        # pass i (side stream) may still be running while pass i+1 hashes: target hashes, the (trusted, target) header
        # hashes, and a never-rewritten copy of the owned ranges' contexts.
        self._parity = 0
        self._target_hashes_pp = [_u8(R * 32, d), _u8(R * 32, d)]
        self._skip_hashes_pp = [_u8(R * 2 * 32, d), _u8(R * 2 * 32, d)]
        self._skip_headers_pp = [_u8(R * 2 * 512, d), _u8(R * 2 * 512, d)]   # the (trusted, target) headers the check reads
        self.inputs_consumed = None                # event: this pass no longer reads headers_all (input streaming)
        self._h2d = None                           # (copy stream, pinned host image of headers_all) when inputs are streamed
        self._h2d_done = None
        self._commit_done = [None, None]           # event per parity: the side stream finished the check that used it
        self.skip_ranges_side = _u8(R * 80, d)
        self.defer_commit_wait = False             # PipelinedEngines: do not join the side stream at the end of a pass
        self.n_map_el = RT * jc * int(self.ml["n_elements"])
        self.n_red_local_el = n_local_nodes * int(self.rl["n_elements"])
        self.n_red_top_el = R * max(world - 1, 0) * int(self.rl["n_elements"])
        self.placement_probe = None
        if with_witness:
            self.witness_map = self._place_witness(self.n_map_el + 2)
            self.witness_red_local = torch.zeros(self.n_red_local_el + 2, dtype=torch.int64, device=d)
            self.witness_red_top = torch.zeros(self.n_red_top_el + 2, dtype=torch.int64, device=d)
        self.events = None
        self.side = torch.cuda.Stream(device=d)
        # which phase the commit side stream starts beside.  Measured on ONE engine object (same allocations, interleaved
        # rounds, tools/exp_prio.py): no commit 7.59 ms/step; beside the hashing 8.58 (generic P7) / 8.29 (keyed);
        # beside the expansion 8.05 (generic) / 7.80 (keyed).  The ALU-bound hashing phase has no spare issue slots,
        # the HBM-bound expansion does — once the field multiplication stopped passing operands through scratch
        # memory (fe25519.h), which used to queue every multiplication behind the expansion's stores.
        self.commit_with = os.environ.get("BSX_COMMIT_WITH", "expand")
        # P7 form: "keyed" rebuilds the per-validator tables from range 0's validator slots every step (nothing is carried
        # between steps) and verifies all R commits against them; a slot whose key differs falls back inside the kernel.
        self.ed_path = os.environ.get("BSX_ED_PATH", "keyed" if R >= 8 else "generic")
        if self.ed_path not in ("keyed", "generic"):
            raise ValueError(f"BSX_ED_PATH={self.ed_path!r}")
        self.keytable = _u8(int(self.L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), d) if self.ed_path == "keyed" else None

    def _place_witness(self, n_el):
        """Allocate the expanded-witness buffer of the map jobs (29.5 GB for 256 x header_range_2048): ONE allocation through
        bsx_dev_alloc (HIP virtual-memory API).  Root cause of round 1's "placement" spread: the store bandwidth of a
        multi-GB buffer depends on where its physical pages lie — hipMalloc'ed buffers of one process ran the same store
        sweep at 5.5-6.6 TB/s, slices of one big arena at 5.4-6.2 TB/s reproducibly by offset, hipMemCreate-backed ones at
        6.0-6.25 TB/s every time (tools/exp_vmm.hip).  BSX_WITNESS_ALLOC=torch falls back to the caching allocator."""
        mode = os.environ.get("BSX_WITNESS_ALLOC", "vmm")
        if mode == "vmm" and n_el * 8 >= (64 << 20):
            self._witness_block = _lib.DeviceBuffer(n_el, self.dev.index if self.dev.index is not None else 0)
            self.placement_probe = {"allocator": "bsx_dev_alloc (hipMemCreate + hipMemMap, one handle)", "candidates": 1}
            return self._witness_block.tensor()
        self.placement_probe = {"allocator": "torch caching allocator (hipMalloc)", "candidates": 1}
        return torch.zeros(n_el, dtype=torch.int64, device=self.dev)

    # ------------------------------------------------------------------ data
    def upload(self, headers_slice, ranges, latest, skip_headers=None, skip_ranges=None, validators=None, trusted=None):
        """headers_slice: [RT, hpr] HEADER (heights S_r + hfr ..); ranges: [RT] SHARED_CTX; latest: [RT] u64.
        Owned ranges: skip_headers [R, 2] HEADER (trusted, target), skip_ranges [R] SHARED_CTX, validators/trusted
        [R, V] VALIDATOR."""
        def put(dst, arr):
            a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
            assert a.size <= dst.numel(), (a.size, dst.numel())
            dst[:a.size].copy_(torch.from_numpy(a), non_blocking=False)
        put(self.headers, headers_slice)
        put(self.ranges, ranges)
        put(self.latest, np.ascontiguousarray(latest, np.uint64))
        if self.with_commit:
            put(self.skip_headers, skip_headers)
            put(self.skip_ranges, skip_ranges)
            put(self.skip_ranges_side, skip_ranges)
            put(self.validators, validators)
            put(self.trusted, trusted)
        torch.cuda.synchronize(self.dev)

    def upload_workload(self, w, sel=None):
        """Convenience for a synth.Workload.  sel: the RT workload ranges this engine touches, ordered [rank][k]
        (default: all of them, single engine); the block `self.rank` of that list is the owned set."""
        sel = np.arange(w.R) if sel is None else np.asarray(sel)
        assert sel.size == self.RT and w.J == self.J and w.B == self.B and w.v_max == self.V
        lo = self.hfr
        hs = w.headers[sel][:, lo:lo + self.hpr]
        own = sel[self.rank * self.R:(self.rank + 1) * self.R]
        sk = np.stack([w.headers[own, 0], w.headers[own, w.n_blocks]], axis=1)
        self.upload(hs, w.ranges[sel], w.latest[sel], sk, w.ranges[own], w.validators[own], w.trusted[own])

    def enable_input_streaming(self, host_image=None):
        """Stream the NEXT pass's headers from pinned host memory while this pass computes (what a caller that does not
        keep its inputs in HBM sees): stream_inputs() enqueues one H2D copy of the whole header block on a copy stream
        as soon as the current pass has consumed the buffer (header hashing + hint assembly, early in the pass), the
        next step_local waits for it.  host_image: pinned uint8 tensor, default = a pinned copy of the resident block."""
        if host_image is None:
            host_image = torch.empty(self.headers_all.numel(), dtype=torch.uint8, pin_memory=True)
            host_image.copy_(self.headers_all)
            torch.cuda.synchronize(self.dev)
        assert host_image.is_pinned() and host_image.numel() == self.headers_all.numel()
        self._h2d = (torch.cuda.Stream(device=self.dev), host_image)

    def stream_inputs(self):
        if self._h2d is None:
            return
        s, img = self._h2d
        if self.inputs_consumed is not None:
            s.wait_event(self.inputs_consumed)
        with torch.cuda.stream(s):
            self.headers_all.copy_(img, non_blocking=True)
            self._h2d_done = torch.cuda.Event()
            self._h2d_done.record(s)

    # ------------------------------------------------------------------ one pass
    def _st(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def step_local(self, time_kernels=False):
        """Stages 1-5 + local fold: everything before the cross-GPU exchange.  The commit verification of the owned
        ranges (stage 3) runs on a side stream: challenges + per-validator tables from here (beside the hashing), the
        signature checks, tallies and skip conditions from launch_verify (beside an expansion)."""
        L, ctx, dp, chk = self.L, self.ctx, _lib.dp, _lib.check
        B, jc, RT, R, V = self.B, self.jc, self.RT, self.R, self.V
        main = torch.cuda.current_stream(self.dev)
        st = self._st()
        ev = self.events if time_kernels else None
        if self._h2d_done is not None:             # streamed inputs: this pass's headers arrive on the copy stream
            main.wait_event(self._h2d_done)
            self._h2d_done = None
        self.status.zero_()
        commit = self.with_commit and R and self.nh_all > RT * self.hpr
        if commit and self.commit_with != "hash":
            # challenges + per-validator tables need nothing from this pass: start them right away beside the hashing
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._commit(self._st(), "prep")
        chk(L.bsx_dev_header_merkle(ctx, st, dp(self.headers_all), C.c_uint64(self.nh_all if commit else RT * self.hpr),
                                    dp(self.hashes_all), dp(self.dh_aunts), dp(self.lb_aunts), dp(self.paths), dp(self.status)))
        self.merkle_done = torch.cuda.Event()
        self.merkle_done.record(main)
        if commit:
            self._parity ^= 1
            done = self._commit_done[self._parity]
            if done is not None:                   # the check two passes ago read this parity's buffers
                main.wait_event(done)
            chk(L.bsx_dev_fill_end_hash(ctx, st, C.c_uint32(R), dp(self.skip_ranges), dp(self.skip_hashes), C.c_uint64(2),
                                        dp(self.target_idx), dp(self.target_hashes), dp(self._skip_hashes_pp[self._parity])))
            if self._h2d is not None:
                # streamed inputs: headers_all is overwritten early in the next pass, the deferred commit check keeps reading the
                # (trusted, target) headers -> private copy per parity (1 KB per range, d2d; 0.5 ms when queued behind the
                # expansion's stores, hence only when needed)
                self._skip_headers_pp[self._parity][:R * 1024].copy_(self.skip_headers[:R * 1024], non_blocking=True)
            self.fill_done = torch.cuda.Event()
            self.fill_done.record(main)
            if self.commit_with == "hash":
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self._commit(self._st(), "all")
        if self.expand_done is not None:           # the previous pass's expansion (on the shared expansion stream) reads `compact`
            main.wait_event(self.expand_done)
            self.expand_done = None
        chk(L.bsx_dev_assemble_inputs(ctx, st, C.c_uint32(RT), C.c_uint32(self.J), C.c_uint32(B), C.c_uint32(self.jf),
                                      C.c_uint32(jc), C.c_uint32(B), dp(self.ranges), dp(self.latest), dp(self.headers),
                                      C.c_uint64(self.hpr), C.c_uint64(self.hfr), dp(self.hashes), dp(self.dh_aunts),
                                      dp(self.lb_aunts), dp(self.compact), dp(self.status[1:]), dp(self.paths)))
        self.inputs_consumed = torch.cuda.Event()
        self.inputs_consumed.record(main)          # headers_all may be overwritten from here on (stream_inputs)
        if ev:
            ev[0].record(main)
        chk(L.bsx_dev_prove_subchain(ctx, st, C.c_uint32(RT), C.c_uint32(B), C.c_uint32(jc), dp(self.ranges), dp(self.compact),
                                     dp(self.records), C.c_uint32(self.subchain_flags)))
        if ev:
            ev[1].record(main)
        chk(L.bsx_dev_reduce(ctx, st, C.c_uint32(RT), C.c_uint32(jc), dp(self.records), dp(self.partial),
                             dp(self.red_compact_local) if jc > 1 else None))

    def _commit(self, st, part="all"):
        """Stage 3 for the owned ranges on stream `st` (builder.skip, header_range.rs:42-48).
        part "prep": SHA-512 challenges + per-validator tables (small, memory-latency sensitive: 1.5 ms + 1.7 ms when
        their loads queue behind the expansion's stores, 0.04 + 0.5 ms otherwise) — run beside the hashing;
        part "verify": the signature checks, tallies and skip conditions (ALU work) — run beside the expansion."""
        L, ctx, dp, chk = self.L, self.ctx, _lib.dp, _lib.check
        R, V = self.R, self.V
        n = R * V
        if part in ("all", "prep"):
            chk(L.bsx_dev_sha512_challenge(ctx, st, dp(self.validators), C.c_uint64(n), dp(self.h), None))
            if self.ed_path == "keyed":
                chk(L.bsx_dev_ed25519_keytable(ctx, st, dp(self.validators), C.c_uint32(V), dp(self.keytable)))
        if part == "prep":
            return
        if self.ed_path == "keyed":
            chk(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(self.validators), dp(self.h), C.c_uint64(n), C.c_uint32(V),
                                               dp(self.keytable), C.c_uint32(V), dp(self.ok), None))
        else:
            chk(L.bsx_dev_ed25519_verify(ctx, st, dp(self.validators), dp(self.h), C.c_uint64(n), dp(self.ok)))
        chk(L.bsx_dev_commit_tally(ctx, st, dp(self.trusted), C.c_uint32(R), C.c_uint32(V), None, None, dp(self.trusted_res)))
        chk(L.bsx_dev_commit_tally(ctx, st, dp(self.validators), C.c_uint32(R), C.c_uint32(V), dp(self.target_hashes), dp(self.ok),
                                   dp(self.commit_res)))
        chk(L.bsx_dev_skip_check(ctx, st, C.c_uint32(R), C.c_uint32(V), dp(self.skip_ranges_side),
                                 dp(self._skip_headers_pp[self._parity] if self._h2d is not None else self.skip_headers),
                                 C.c_uint64(2), dp(self._skip_hashes_pp[self._parity]), dp(self.validators), dp(self.trusted),
                                 dp(self.ok), dp(self.commit_res), dp(self.trusted_res), dp(self.skip_status), None,
                                 dp(self.target_idx), _lib.p(self.chain_id) if self.chain_id.size else None,
                                 C.c_uint32(self.chain_id.size)))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._commit_done[self._parity] = ev

    @property
    def target_hashes(self):
        return self._target_hashes_pp[self._parity]

    def step_exchange(self, gathered=None):
        """Stage 6: the one collective.  Single GPU: the local fold already is the range result.
        gathered (tests only): a [world, RT, 128] uint8 tensor standing in for the all-gather result, so that all
        ranks' engines can be exercised on ONE GPU without a process group."""
        if self.world == 1:
            return self.partial
        if gathered is None:
            gathered = all_gather_records(self.partial, self.world, self.RT, self.gathered)
        return self._top_fold(gathered)

    def _top_fold(self, gathered):
        # top fold straight from the all-gather layout [rank][range]: record k of owned range r = gathered[k][rank*R + r]
        own = gathered.view(-1)[self.rank * self.R * 128:]
        L, ctx, st, dp, chk = self.L, self.ctx, self._st(), _lib.dp, _lib.check
        chk(L.bsx_dev_reduce_strided(ctx, st, C.c_uint32(self.R), C.c_uint32(self.world), dp(own), C.c_uint64(1),
                                     C.c_uint64(self.RT), dp(self.results), dp(self.red_compact_top)))
        return self.results

    def step_exchange_begin(self):
        """Start the collective without waiting for it (N > 1): the map-job expansion needs nothing from it, so
        PipelinedEngines enqueues that expansion next and finishes the exchange (top fold, finalize) behind it —
        the all-gather's latency, inflated while every GPU's HBM is saturated, then hides beside this chunk's own
        expansion instead of delaying it."""
        self._gather_pending = None
        if self.world > 1:
            self._gather_pending = all_gather_records(self.partial, self.world, self.RT, self.gathered, async_op=True)

    def step_exchange_end(self):
        if self.world == 1:
            return self.partial
        gathered, work = self._gather_pending
        if work is not None:
            work.wait()                 # stream-level wait on the collective (the host does not block for NCCL)
        self._gather_pending = None
        return self._top_fold(gathered)

    def launch_verify(self, after_event=None):
        """Signature checks, tallies and skip conditions of the current pass on the side stream ("expand" placement).
        They need the target hashes (fill_end_hash) and the prep part already queued on the side stream; after_event
        (optional) delays them further — PipelinedEngines passes the other chunk's header-hashing event, so that this
        ALU work runs beside that chunk's memory-leaning kernels instead of beside its k_header_merkle."""
        if not (self.with_commit and self.R and self.commit_with == "expand"):
            return
        self.side.wait_event(self.fill_done)
        if after_event is not None:
            self.side.wait_event(after_event)
        with torch.cuda.stream(self.side):
            self._commit(self._st(), "verify")

    def step_final(self, result_records, time_kernels=False, before_expand=None, launch_verify=True, after_expand=None,
                   expand_stream=None):
        """finalize + (commit verification on the side stream) + witness expansion.  before_expand: hook called right
        before the expansion is enqueued (PipelinedEngines waits for the other chunk's expansion there, so that the tiny
        finalize kernel and the side-stream launch do not sit between two expansions); after_expand: hook called right
        behind the map-job expansion (PipelinedEngines releases the other chunk's expansion there).
        result_records None = the exchange was only begun (step_exchange_begin): top fold, finalize and the top
        reduce nodes' expansion then run behind the map-job expansion.
        expand_stream: launch the map-job / local reduce-node expansions there instead of on the current stream
        (PipelinedEngines: ONE stream for the expansions of all chunks, so that consecutive expansions are consecutive
        packets of one hardware queue instead of an event hand-over between two: the hand-over left HBM idle for 50-80 us
        per expansion); this chunk's next hint then waits for `expand_done`."""
        L, ctx, st, dp, chk = self.L, self.ctx, self._st(), _lib.dp, _lib.check
        ev = self.events if time_kernels else None
        own_ranges = self.skip_ranges if self.with_commit else self.ranges[self.rank * self.R * 80:]

        def finalize(records):
            chk(L.bsx_dev_finalize(ctx, st, C.c_uint32(self.R), C.c_uint32(self.J), C.c_uint32(self.B), dp(own_ranges),
                                   dp(records), dp(self.target_hashes) if self.with_commit else None, dp(self.output64),
                                   dp(self.range_status)))
        late = result_records is None          # exchange still in flight (step_exchange_begin): finish it behind the expansion
        if not late:
            finalize(result_records)
        if launch_verify:
            # integer-ALU work: start it beside the HBM-bound expansion (i.e. once finalize is done), not beside the hashing
            fin = torch.cuda.Event()
            fin.record(torch.cuda.current_stream(self.dev))
            self.launch_verify(after_event=fin)
        if before_expand is not None:
            before_expand()
        if self.with_witness:
            xs, stx = torch.cuda.current_stream(self.dev), st
            if expand_stream is not None:
                ready = torch.cuda.Event()
                ready.record(xs)                       # the compact witnesses (hint, prove_subchain, reduce) are complete
                expand_stream.wait_event(ready)
                xs, stx = expand_stream, C.c_void_p(expand_stream.cuda_stream)
            if ev:
                ev[2].record(xs)
            chk(L.bsx_dev_expand_witness(ctx, stx, _lib.p(self._ml), C.c_uint32(self.RT * self.jc), dp(self.compact),
                                         dp(self.witness_map)))
            if ev:
                ev[3].record(xs)
            if self.jc > 1:
                chk(L.bsx_dev_expand_witness(ctx, stx, _lib.p(self._rl), C.c_uint32(self.RT * (self.jc - 1)),
                                             dp(self.red_compact_local), dp(self.witness_red_local)))
            if expand_stream is not None:
                self.expand_done = torch.cuda.Event()
                self.expand_done.record(xs)
        if after_expand is not None:
            after_expand()
        if late:
            finalize(self.step_exchange_end())
        if self.with_witness:
            if self.world > 1:
                chk(L.bsx_dev_expand_witness(ctx, st, _lib.p(self._rl), C.c_uint32(self.R * (self.world - 1)),
                                             dp(self.red_compact_top), dp(self.witness_red_top)))

    def step(self, time_kernels=False):
        self.step_local(time_kernels)
        self.stream_inputs()
        res = self.step_exchange()
        self.step_final(res, time_kernels)
        self.join_commit()

    def join_commit(self):
        """Make the current stream wait for the commit check on the side stream."""
        if self.with_commit and self.R:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)

    def enable_timing(self):
        self.events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    # ------------------------------------------------------------------ results
    def download(self):
        torch.cuda.synchronize(self.dev)
        out = dict(
            output64=self.output64[:self.R * 64].cpu().numpy().reshape(self.R, 64),
            range_status=self.range_status[:self.R].cpu().numpy().astype(np.uint32),
            header_status=int(self.status[0].item()), assemble_status=int(self.status[1].item()),
            records=self.records[:self.RT * self.jc * 128].cpu().numpy().view(T.SUBCHAIN).reshape(self.RT, self.jc),
        )
        if self.with_commit:
            out["skip_status"] = self.skip_status[:self.R].cpu().numpy().astype(np.uint32)
            out["commit"] = self.commit_res[:self.R * 96].cpu().numpy().view(T.COMMIT_RESULT)
        return out

    def witness_numpy(self):
        torch.cuda.synchronize(self.dev)
        m = self.witness_map[:self.n_map_el].cpu().numpy().view(np.uint64)
        rl = self.witness_red_local[:self.n_red_local_el].cpu().numpy().view(np.uint64)
        rt = self.witness_red_top[:self.n_red_top_el].cpu().numpy().view(np.uint64)
        return m, rl, rt


class PipelinedEngines:
    """E engines, each over 1/E of the step's ranges on its own HIP stream.  The SHA kernels are integer-ALU bound and
    the witness expansion is HBM-write bound, so running chunk e+1's hashing beside chunk e's expansion overlaps the
    two resources; consecutive steps pipeline the same way (each engine's buffers are only touched on its own stream)."""

    def __init__(self, nb_map_jobs, batch_size, v_max, n_ranges_local, n_engines=2, rank=0, world=1, device=None, **kw):
        assert n_ranges_local % n_engines == 0
        self.E, self.Rc, self.R, self.rank, self.world = n_engines, n_ranges_local // n_engines, n_ranges_local, rank, world
        self.engines = [HeaderRangeEngine(nb_map_jobs, batch_size, v_max, self.Rc, rank=rank, world=world, device=device, **kw)
                        for _ in range(n_engines)]
        self.dev = self.engines[0].dev
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(n_engines)]
        self._hash_token = None
        self._expand_token = None
        self._pending_verify = None
        self.verify_after_merkle = os.environ.get("BSX_VERIFY_AFTER_MERKLE", "1") == "1"
        # BSX_EXPAND_STREAM=1 (experiment, off): one stream for the expansions of all chunks (step_final) instead of the chunks'
        # own streams with an event token between them: +1 % per step at header_range_2048, -6 % at header_range_1024
        self.xstream = (torch.cuda.Stream(device=self.dev) if n_engines > 1 and self.engines[0].with_witness
                        and os.environ.get("BSX_EXPAND_STREAM", "0") == "1" else None)
        # k_header_merkle alone fills the register file (4 waves x 128 VGPRs per SIMD); beside an expansion it is held to
        # 2 workgroups per CU so that the expansion's waves keep half of it (bsx.h BSX_TUNE_MERKLE_WORKGROUPS): +2 % per step
        e0 = self.engines[0]
        if e0.with_witness and n_engines > 1:
            for e in self.engines:           # the one-launch prove_subchain holds 4 x 128 registers per SIMD: same trade
                e.subchain_flags |= 2 if e.fused_hint else 0
        _lib.check(e0.L.bsx_set_tuning(e0.ctx, C.c_uint32(T.TUNE_MERKLE_WORKGROUPS),
                                       C.c_uint64(2 * torch.cuda.get_device_properties(self.dev).multi_processor_count
                                                  if e0.with_witness and n_engines > 1 else 0)))

    def sel(self, e):
        return np.concatenate([np.arange(g * self.R + e * self.Rc, g * self.R + (e + 1) * self.Rc) for g in range(self.world)])

    def upload_workload(self, w):
        assert w.R == self.R * self.world
        for e, eng in enumerate(self.engines):
            eng.upload_workload(w, self.sel(e))

    def step(self, time_kernels=False, events=None):
        """One pass over all chunks.  Two tokens keep the chunks in complementary phases: only one chunk hashes at a time
        and only one expands at a time, so chunk e+1's (ALU-bound) hashing always runs beside chunk e's (HBM-bound)
        expansion — without the tokens the streams drift into the same phase and the overlap is lost."""
        for e, (eng, s) in enumerate(zip(self.engines, self.streams)):
            with torch.cuda.stream(s):
                if events is not None:
                    eng.events = events[e]
                if self.E > 1 and self._hash_token is not None:
                    s.wait_event(self._hash_token)
                eng.step_local(time_kernels)
                eng.stream_inputs()                # no-op unless enable_input_streaming(): next pass's headers, H2D
                if self._pending_verify is not None:
                    # the previous chunk's signature checks: enqueued now so that they can wait for THIS chunk's
                    # k_header_merkle (both are integer-ALU bound; the rest of this chunk's hashing phase leans on memory)
                    self._pending_verify.launch_verify(after_event=eng.merkle_done)
                    self._pending_verify = None
                if self.world > 1:
                    eng.step_exchange_begin()      # collective in flight; finished behind this chunk's expansion
                    res = None
                else:
                    res = eng.step_exchange()
                if self.E > 1:
                    self._hash_token = torch.cuda.Event()
                    self._hash_token.record(s)
                tok = self._expand_token if self.E > 1 and self.xstream is None else None
                defer = self.verify_after_merkle and self.E > 1
                def release(s=s):
                    if self.E > 1 and self.xstream is None:
                        self._expand_token = torch.cuda.Event()
                        self._expand_token.record(s)
                eng.step_final(res, time_kernels, before_expand=(lambda s=s, tok=tok: s.wait_event(tok)) if tok is not None else None,
                               launch_verify=not defer, after_expand=release, expand_stream=self.xstream)
                if defer:
                    self._pending_verify = eng
                # the commit check is NOT joined here: its inputs are double-buffered by pass parity (HeaderRangeEngine), so
                # it may run on into the chunk's next pass; join() / download() wait for it

    def join(self):
        if self._pending_verify is not None:          # no later chunk to wait for: launch the deferred checks now
            self._pending_verify.launch_verify()
            self._pending_verify = None
        cur = torch.cuda.current_stream(self.dev)
        if self.xstream is not None:
            cur.wait_stream(self.xstream)
        for s, eng in zip(self.streams, self.engines):
            cur.wait_stream(s)
            if eng.with_commit and eng.R:
                cur.wait_stream(eng.side)

    def download(self):
        self.join()
        outs = [eng.download() for eng in self.engines]
        merged = {}
        for k in outs[0]:
            v0 = outs[0][k]
            merged[k] = np.concatenate([o[k] for o in outs]) if isinstance(v0, np.ndarray) else max(o[k] for o in outs)
        return merged


class AlternatingPipelines:
    """K PipelinedEngines over the SAME ranges, stepped in turn: software pipelining ACROSS steps (step i + 1 starts on its own
    buffers while step i's chain of small kernels drains).  The form for the compact-only path, whose step is a serial chain
    (header hashing, hint, prove_subchain, reduce, finalize) with nothing HBM-bound to hide behind: 322 -> 343-392 M headers/s
    with K = 2.  With the witness it does not pay (the expansions of two steps share HBM) and doubles the 29 GB image."""

    def __init__(self, k, *args, **kw):
        self.sets = [PipelinedEngines(*args, **kw) for _ in range(k)]
        self.K, self.i = k, 0
        s0 = self.sets[0]
        self.engines, self.dev, self.R, self.E = s0.engines, s0.dev, s0.R, s0.E

    def sel(self, e):
        return self.sets[0].sel(e)

    def upload_workload(self, w):
        for s in self.sets:
            s.upload_workload(w)

    def step(self, time_kernels=False, events=None):
        self.sets[self.i % self.K].step(time_kernels, events)
        self.i += 1

    def join(self):
        for s in self.sets:
            s.join()

    def download(self):
        """Results of the most recent step; every set that has stepped must hold the same public outputs."""
        self.join()
        outs = [s.download() for s in self.sets[:min(self.i, self.K)]]
        for o in outs[1:]:
            assert (o["output64"] == outs[0]["output64"]).all() and (o["range_status"] == outs[0]["range_status"]).all()
        return outs[(self.i - 1) % self.K if self.i else 0]

