"""Python face of the batched header_range pipeline — a THIN ctypes wrapper over `bsx_pipeline_*` (include/bsx.h,
csrc/pipeline.hip).  The pipeline object in the C library owns the device buffers, the HIP streams and events, the
parity double-buffers and every launch-form decision; nothing here enqueues a kernel or reads an environment variable.
What stays in Python: numpy marshalling of the inputs / results, zero-copy views of the library's device buffers for the
tests, and — across GPUs — the one collective (torch.distributed over RCCL), handed to the library as its all-gather
callback.

Step over R ranges of J map jobs x B headers (reference shapes 32x32 / 32x64, bin/header_range_{1024,2048}.rs:6-17):

  1 header_merkle     tendermint Header::hash + inclusion proofs for every supplied header   (input.rs:175-195,250-261)
  2 fill_end_hash     ctx.end_header_hash := target header hash (output of builder.skip)     (header_range.rs:42-55)
  3 commit            SHA-512 challenge -> Ed25519 -> tallies/validator hashes -> skip check  (header_range.rs:42-48)
  4 assemble_inputs   the hint of every map job                                               (data_commitment.rs:22-44)
  5 prove_subchain    map stage                                                               (builder.rs:150-271,305-336)
  6 reduce            local fold of this device's jobs, [all-gather across GPUs], top fold    (builder.rs:337-395)
  7 finalize          range check, final asserts, 64-byte public output                       (builder.rs:292-297,400-406)
  8 expand_witness    compact witness -> Goldilocks elements (map jobs + reduce nodes) and/or Poseidon Merkle caps

Multi-GPU: rank g owns map jobs [g*J/N, (g+1)*J/N) of EVERY range of the global batch (N*R ranges, so the
per-GPU slot count is the same as at N = 1: weak scaling), folds them locally, and ONE all-gather of a 128-byte
record per (range, rank) replaces the reference's map->reduce hand-off; the owner of a range (range index // R) does
the last log2(N) reduce levels, the final assertions and that range's commit verification.
"""
import ctypes as C
import traceback

import numpy as np

from . import _lib
from . import types as T

# bsx.h
PIPE_WITNESS, PIPE_COMMIT, PIPE_CAPS, PIPE_ED_GENERIC, PIPE_COMMIT_BESIDE_HASH, PIPE_RECOMPUTE_PATHS, PIPE_NO_UNITS = 1, 2, 4, 8, 16, 32, 64
(BUF_WITNESS_MAP, BUF_WITNESS_REDUCE_LOCAL, BUF_WITNESS_REDUCE_TOP, BUF_COMPACT, BUF_TREES, BUF_PARTIAL, BUF_HEADERS, BUF_RECORDS,
 BUF_GATHERED, BUF_REDUCE_COMPACT_LOCAL, BUF_HASHES, BUF_DH_AUNTS, BUF_LB_AUNTS, BUF_PATHS, BUF_RANGES, BUF_WITNESS_COMMIT,
 BUF_WITNESS_SKIP, BUF_COMPACT_COMMIT, BUF_COMPACT_SKIP, BUF_TREES_COMMIT, BUF_TREES_SKIP) = range(21)


class _Config(C.Structure):
    _fields_ = [("nb_map_jobs", C.c_uint32), ("batch_size", C.c_uint32), ("v_max", C.c_uint32), ("n_ranges", C.c_uint32),
                ("n_chunks", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32), ("flags", C.c_uint32),
                ("leaf_len", C.c_uint32), ("cap_height", C.c_uint32), ("chain_id_len", C.c_uint32), ("chain_id", C.c_uint8 * 52),
                ("tune_merkle_workgroups", C.c_uint32), ("tune_subchain", C.c_uint32), ("n_sets", C.c_uint32), ("_reserved", C.c_uint32)]


class _Inputs(C.Structure):
    _fields_ = [("headers", C.c_void_p), ("headers_per_range", C.c_uint64), ("ranges", C.c_void_p), ("latest", C.c_void_p),
                ("target_validators", C.c_void_p), ("trusted_validators", C.c_void_p)]


class _Results(C.Structure):
    _fields_ = [("output64", C.c_void_p), ("range_status", C.c_void_p), ("skip_status", C.c_void_p), ("commit", C.c_void_p),
                ("records", C.c_void_p), ("header_status", C.c_uint32), ("assemble_status", C.c_uint32)]


class _Timing(C.Structure):
    _fields_ = [("prove_subchain_ms", C.c_double), ("expand_map_ms", C.c_double), ("caps_ms", C.c_double), ("launches", C.c_uint32),
                ("exchanges", C.c_uint32), ("allgather_ms_avg", C.c_double), ("allgather_ms_min", C.c_double), ("allgather_ms_median", C.c_double),
                ("allgather_ms_max", C.c_double)]


class _Autotune(C.Structure):
    _fields_ = [("n_trials", C.c_uint32), ("best_trial", C.c_uint32), ("steps_per_trial", C.c_uint32), ("hw_queues", C.c_uint32),
                ("initial_ms", C.c_double), ("best_ms", C.c_double), ("worst_ms", C.c_double), ("assignment", C.c_uint32 * 16)]


assert C.sizeof(_Autotune) == 104
assert C.sizeof(_Config) == 112 and C.sizeof(_Inputs) == 48 and C.sizeof(_Results) == 48 and C.sizeof(_Timing) == 64
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


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
    import torch
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
    device path folds the gathered layout in place with a strided reduce)."""
    import torch
    RT = world * n_ranges_local
    gathered = all_gather_records(partial, world, RT, out_gathered)
    g = gathered.view(world, RT, 128)
    own = g[:, rank * n_ranges_local:(rank + 1) * n_ranges_local, :]            # [rank, owned range, 128]
    top = out_top[:n_ranges_local * world * 128] if out_top is not None else torch.empty(n_ranges_local * world * 128, dtype=torch.uint8, device=gathered.device)
    top.view(n_ranges_local, world, 128).copy_(own.transpose(0, 1))
    return top


class _DevView:
    """A library-owned device buffer as a zero-copy torch tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _view(ptr, nbytes, dev, i64=False):
    import torch
    if not ptr or not nbytes:
        return torch.zeros(0, dtype=torch.int64 if i64 else torch.uint8, device=dev)
    return torch.as_tensor(_DevView(ptr, nbytes, "<i8", 8) if i64 else _DevView(ptr, nbytes), device=dev)


def torch_allgather(dev, world):
    """The all-gather callback over torch.distributed: RCCL ("nccl") enqueues on its own stream behind `stream` and makes
    `stream` wait for it (no host block); gloo (tests: ranks sharing one GPU) stages through host memory."""
    import torch
    import torch.distributed as dist

    def fn(send, recv, stream):
        ext = torch.cuda.ExternalStream(stream, device=dev)
        if dist.get_backend() == "gloo":
            ext.synchronize()
            g_cpu = torch.empty(recv.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(g_cpu, send.cpu())
            with torch.cuda.stream(ext):
                recv.copy_(g_cpu)
            ext.synchronize()
        else:
            with torch.cuda.stream(ext):
                dist.all_gather_into_tensor(recv, send)
    return fn


def c_rccl_comm(ctx, rank, world):
    """An RCCL communicator made through the C ABI (bsx_rccl_get_unique_id / bsx_rccl_comm_init_rank, include/bsx.h): rank 0's
    128-byte unique id travels over the already initialised torch.distributed group — control plane only; the collective of the
    data path is then ncclAllGather called by the library itself (bsx_pipeline_set_rccl), no Python in the loop.  Collective."""
    L = _lib.lib()
    idb = np.zeros(128, np.uint8)
    if rank == 0:
        _lib.check(L.bsx_rccl_get_unique_id(_lib.p(idb)))
    if world > 1:
        import torch.distributed as dist
        obj = [idb.tobytes()]
        dist.broadcast_object_list(obj, src=0)
        idb = np.frombuffer(obj[0], np.uint8).copy()
    comm = C.c_void_p()
    _lib.check(L.bsx_rccl_comm_init_rank(ctx, C.c_uint32(world), _lib.p(idb), C.c_uint32(rank), C.byref(comm)))
    return comm


class Pipeline:
    """bsx_pipeline: `n_ranges_local` header_range instances per step on this rank, cut into `n_chunks` chunks."""

    def __init__(self, nb_map_jobs, batch_size, v_max, n_ranges_local, n_chunks=1, rank=0, world=1, device=None, with_witness=True,
                 with_commit=True, with_caps=False, chain_id=b"celestia", ed_path=None, commit_with="expand", fused_hint=True,
                 leaf_len=0, cap_height=0, merkle_workgroups=0, subchain_form=0, n_sets=1, units=True):
        import torch
        self.J, self.B, self.V = nb_map_jobs, batch_size, v_max
        self.R, self.E, self.Rc = n_ranges_local, n_chunks, n_ranges_local // n_chunks
        self.rank, self.world = rank, world
        self.RT = self.Rc * world                   # ranges per chunk
        self.jc = nb_map_jobs // world if world else 0            # the library validates the shape (bsx_pipeline_create)
        self.jf = rank * self.jc
        self.with_witness, self.with_commit, self.with_caps = with_witness, with_commit, with_caps
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dev_index = self.dev.index if self.dev.index is not None else 0
        self.ctx = _lib.context(self.dev_index)
        self.L = _lib.lib()
        self.ml, self.rl = T.map_layout(batch_size), T.reduce_layout()
        self._ml = np.array(self.ml).reshape(1)
        self._rl = np.array(self.rl).reshape(1)
        if ed_path not in (None, "keyed", "generic"):
            raise ValueError(f"ed_path={ed_path!r}")
        if commit_with not in ("expand", "hash"):
            raise ValueError(f"commit_with={commit_with!r}")
        flags = (PIPE_WITNESS if with_witness else 0) | (PIPE_COMMIT if with_commit else 0) | (PIPE_CAPS if with_caps else 0)
        flags |= PIPE_ED_GENERIC if ed_path == "generic" else 0
        flags |= PIPE_COMMIT_BESIDE_HASH if commit_with == "hash" else 0
        flags |= 0 if fused_hint else PIPE_RECOMPUTE_PATHS
        flags |= 0 if units else PIPE_NO_UNITS
        self.units = bool(units and with_commit and (with_witness or with_caps))
        self.fused_hint = fused_hint
        self._leaf_len = leaf_len or 135
        self.ed_path = "generic" if (ed_path == "generic" or self.Rc < 8) else "keyed"
        self.commit_with = commit_with
        cid = bytes(chain_id)
        cfg = _Config(nb_map_jobs, batch_size, v_max, n_ranges_local, n_chunks, rank, world, flags, leaf_len, cap_height, len(cid),
                      (C.c_uint8 * 52)(*cid[:52]), merkle_workgroups, subchain_form, n_sets, 0)
        self.K = n_sets
        self._h = C.c_void_p()
        _lib.check(self.L.bsx_pipeline_create(self.ctx, C.byref(cfg), C.byref(self._h)))
        self._cb = None
        if world > 1:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.set_allgather(torch_allgather(self.dev, world))

    def __del__(self):
        try:
            if self._h:
                self.L.bsx_pipeline_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    close = __del__

    # ------------------------------------------------------------------ multi-GPU exchange
    def set_allgather(self, fn):
        """fn(send, recv, stream): all-gather ordered on the HIP stream `stream` (int); send / recv are uint8 device tensors
        viewing the library's buffers ([RT*128] and [world][RT*128])."""
        dev, world = self.dev, self.world

        def cb(user, d_send, d_recv, nbytes, stream):
            try:
                fn(_view(d_send, nbytes, dev), _view(d_recv, nbytes * world, dev), int(stream or 0))
                return 0
            except Exception:            # noqa: BLE001 — reported through the C return code
                traceback.print_exc()
                return 1
        self._cb = ALLGATHER_FN(cb)      # keep the trampoline alive
        _lib.check(self.L.bsx_pipeline_set_allgather(self._h, self._cb, None))

    def set_rccl(self, comm):
        """bsx_pipeline_set_rccl: the library calls ncclAllGather on `comm` (an ncclComm_t as c_void_p) itself; then
        bsx_pipeline_check_allgather proves the collective end to end (collective call)."""
        _lib.check(self.L.bsx_pipeline_set_rccl(self._h, comm))
        self._cb = None
        _lib.check(self.L.bsx_pipeline_check_allgather(self._h))

    def check_allgather(self):
        _lib.check(self.L.bsx_pipeline_check_allgather(self._h))

    # ------------------------------------------------------------------ data
    def upload(self, headers, ranges, latest, validators=None, trusted=None):
        """headers [world*R, hpr_full] HEADER (height S_r + k), ranges [world*R] SHARED_CTX, latest [world*R] u64,
        validators / trusted [world*R, V] VALIDATOR — global range order r = owner*R + k (bsx.h bsx_pipeline_inputs)."""
        n = self.world * self.R
        h = np.ascontiguousarray(headers, T.HEADER)
        assert h.shape[0] == n, (h.shape, n)
        rg = np.ascontiguousarray(ranges, T.SHARED_CTX).reshape(n)
        la = np.ascontiguousarray(latest, np.uint64).reshape(n)
        tv = np.ascontiguousarray(validators, T.VALIDATOR).reshape(n, self.V) if validators is not None else None
        rv = np.ascontiguousarray(trusted, T.VALIDATOR).reshape(n, self.V) if trusted is not None else None
        inp = _Inputs(h.ctypes.data, h.shape[1], rg.ctypes.data, la.ctypes.data, tv.ctypes.data if tv is not None else None,
                      rv.ctypes.data if rv is not None else None)
        _lib.check(self.L.bsx_pipeline_upload(self._h, C.byref(inp)))

    def upload_workload(self, w, sel=None):
        """A synth.Workload; sel: the world*R workload ranges this pipeline touches in global order (default: all)."""
        sel = np.arange(w.R) if sel is None else np.asarray(sel)
        assert sel.size == self.world * self.R and w.J == self.J and w.B == self.B and w.v_max == self.V
        self.upload(w.headers[sel], w.ranges[sel], w.latest[sel], w.validators[sel], w.trusted[sel])

    def enable_input_streaming(self, on=True):
        _lib.check(self.L.bsx_pipeline_enable_input_streaming(self._h, C.c_int(1 if on else 0)))

    # ------------------------------------------------------------------ run
    def step(self):
        _lib.check(self.L.bsx_pipeline_step(self._h))

    def join(self):
        _lib.check(self.L.bsx_pipeline_join(self._h))

    def autotune(self, steps_per_trial=0):
        """bsx_pipeline_autotune: measure which hardware queues the chunks' streams overlap best on; keeps the fastest."""
        r = _Autotune()
        _lib.check(self.L.bsx_pipeline_autotune(self._h, C.c_uint32(steps_per_trial), C.byref(r)))
        return {"n_trials": r.n_trials, "best_trial": r.best_trial, "steps_per_trial": r.steps_per_trial, "initial_ms": r.initial_ms,
                "best_ms": r.best_ms, "worst_ms": r.worst_ms, "hw_queues": r.hw_queues, "assignment": list(r.assignment)[:2 * self.E * getattr(self, "K", 1)]}

    def set_timing(self, on=True):
        _lib.check(self.L.bsx_pipeline_set_timing(self._h, C.c_int(1 if on else 0)))

    def timing(self):
        t = _Timing()
        _lib.check(self.L.bsx_pipeline_timing2(self._h, C.byref(t), C.c_uint32(C.sizeof(t))))
        return {"prove_subchain_ms": t.prove_subchain_ms, "expand_map_ms": t.expand_map_ms, "caps_ms": t.caps_ms, "launches": t.launches,
                "exchanges": t.exchanges, "allgather_us": {"avg": t.allgather_ms_avg * 1e3, "min": t.allgather_ms_min * 1e3,
                                                           "median": t.allgather_ms_median * 1e3, "max": t.allgather_ms_max * 1e3}}

    # ------------------------------------------------------------------ results
    def download(self):
        R, n = self.R, self.world * self.R
        out = dict(output64=np.zeros((R, 64), np.uint8), range_status=np.zeros(R, np.uint32), skip_status=np.zeros(R, np.uint32),
                   commit=np.zeros(R, T.COMMIT_RESULT), records=np.zeros((n, self.jc), T.SUBCHAIN))
        res = _Results(out["output64"].ctypes.data, out["range_status"].ctypes.data, out["skip_status"].ctypes.data,
                       out["commit"].ctypes.data, out["records"].ctypes.data, 0, 0)
        _lib.check(self.L.bsx_pipeline_get_results(self._h, C.byref(res)))
        out["header_status"], out["assemble_status"] = int(res.header_status), int(res.assemble_status)
        if not self.with_commit:
            del out["skip_status"], out["commit"]
        return out

    def buffer(self, chunk, which, i64=False):
        """Zero-copy torch view of a device buffer of chunk `chunk` (bsx_pipeline_buffer; with buffer sets the index is
        set * n_chunks + chunk)."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _lib.check(self.L.bsx_pipeline_buffer(self._h, C.c_uint32(chunk), C.c_uint32(which), C.byref(ptr), C.byref(n)))
        return _view(ptr.value, n.value, self.dev, i64)

    def witness_numpy(self, chunk=0):
        """(map-job witness, local reduce nodes, top reduce nodes) of a chunk as uint64 arrays (joins first)."""
        self.join()
        return tuple(self.buffer(chunk, b, i64=True).cpu().numpy().view(np.uint64)
                     for b in (BUF_WITNESS_MAP, BUF_WITNESS_REDUCE_LOCAL, BUF_WITNESS_REDUCE_TOP))

    def unit_witness_numpy(self, chunk=0):
        """BSX_PIPE_WITNESS + BSX_PIPE_COMMIT: (COMMIT units [Rc, n_el], SKIP units [Rc, n_el]) of a chunk's owned ranges — the
        variables of builder.skip (include/bsx_layout.h) as uint64 (joins first)."""
        self.join()
        cl, sl = T.commit_layout(self.V), T.skip_layout(self.V)
        c = self.buffer(chunk, BUF_WITNESS_COMMIT, i64=True).cpu().numpy().view(np.uint64).reshape(self.Rc, int(cl["n_elements"]))
        s = self.buffer(chunk, BUF_WITNESS_SKIP, i64=True).cpu().numpy().view(np.uint64).reshape(self.Rc, int(sl["n_elements"]))
        return c, s

    def unit_caps_numpy(self, chunk=0):
        """BSX_PIPE_CAPS + BSX_PIPE_COMMIT: the Poseidon trees [Rc][digests][4] of the COMMIT and of the SKIP units (joins first)."""
        self.join()
        tc = self.buffer(chunk, BUF_TREES_COMMIT, i64=True).cpu().numpy().view(np.uint64).reshape(self.Rc, -1, 4)
        ts = self.buffer(chunk, BUF_TREES_SKIP, i64=True).cpu().numpy().view(np.uint64).reshape(self.Rc, -1, 4)
        return tc, ts

    def caps_numpy(self, chunk=0):
        """BSX_PIPE_CAPS: (trees [jobs][digests][4], caps [jobs][2^cap_height][4]) of a chunk's map jobs (joins first)."""
        self.join()
        t = self.buffer(chunk, BUF_TREES, i64=True).cpu().numpy().view(np.uint64)
        n_jobs = self.RT * self.jc
        t = t.reshape(n_jobs, -1, 4)
        n_leaves = int(self.L.bsx_witness_leaf_count(C.c_uint64(int(self.ml["n_elements"])), C.c_uint32(self.leaf_len)))
        ncap = 2 * n_leaves - t.shape[1]
        return t, t[:, t.shape[1] - ncap:, :]

    @property
    def leaf_len(self):
        return self._leaf_len

    def sel(self, e):
        """Global range indices of chunk e, in the chunk's own order [owner][k]."""
        return np.concatenate([np.arange(g * self.R + e * self.Rc, g * self.R + (e + 1) * self.Rc) for g in range(self.world)])


class HeaderRangeEngine(Pipeline):
    """One chunk (no intra-step pipelining): the form of the small parity cases.  Device buffers are exposed as attributes
    (zero-copy views of the library's allocations) for the tests that read intermediate state."""

    def __init__(self, nb_map_jobs, batch_size, v_max, n_ranges_local, rank=0, world=1, device=None, **kw):
        super().__init__(nb_map_jobs, batch_size, v_max, n_ranges_local, n_chunks=1, rank=rank, world=world, device=device, **kw)

    compact = property(lambda s: s.buffer(0, BUF_COMPACT))
    red_compact_local = property(lambda s: s.buffer(0, BUF_REDUCE_COMPACT_LOCAL))
    witness_map = property(lambda s: s.buffer(0, BUF_WITNESS_MAP, i64=True))
    witness_red_local = property(lambda s: s.buffer(0, BUF_WITNESS_REDUCE_LOCAL, i64=True))
    witness_red_top = property(lambda s: s.buffer(0, BUF_WITNESS_REDUCE_TOP, i64=True))
    partial = property(lambda s: s.buffer(0, BUF_PARTIAL))
    hashes_all = property(lambda s: s.buffer(0, BUF_HASHES))
    dh_aunts = property(lambda s: s.buffer(0, BUF_DH_AUNTS))
    lb_aunts = property(lambda s: s.buffer(0, BUF_LB_AUNTS))
    paths = property(lambda s: s.buffer(0, BUF_PATHS) if s.fused_hint else None)


class PipelinedEngines(Pipeline):
    """n_engines chunks on their own HIP streams (inside the library): the object bench.py times."""

    def __init__(self, nb_map_jobs, batch_size, v_max, n_ranges_local, n_engines=2, rank=0, world=1, device=None, **kw):
        assert n_ranges_local % n_engines == 0
        super().__init__(nb_map_jobs, batch_size, v_max, n_ranges_local, n_chunks=n_engines, rank=rank, world=world, device=device, **kw)


class AlternatingPipelines(Pipeline):
    """K buffer sets inside ONE bsx_pipeline (bsx_pipeline_config.n_sets): step i runs on set i mod K, so step i + 1 starts on
    its own buffers while step i's chain of short kernels drains, and a token lets exactly one header hashing run at a time.
    The form for the compact-only path; with the witness it does not pay (two expansions share HBM, twice the 29 GB image)."""

    def __init__(self, k, nb_map_jobs, batch_size, v_max, n_ranges_local, n_engines=1, rank=0, world=1, device=None, **kw):
        super().__init__(nb_map_jobs, batch_size, v_max, n_ranges_local, n_chunks=n_engines, rank=rank, world=world, device=device, n_sets=k, **kw)


def run_world_on_one_gpu(engines):
    """Tests: every rank's pipeline of an N-GPU configuration on ONE GPU, without a process group.  Pass 1 steps every rank
    with a no-op exchange to obtain its locally folded records; pass 2 repeats the (deterministic) step with an all-gather
    callback that delivers the concatenation of all ranks' records — exactly ncclAllGather's result — to each rank."""
    import torch
    world = len(engines)
    if world == 1:
        engines[0].step()
        engines[0].join()
        return
    for e in engines:
        e.set_allgather(lambda send, recv, stream: None)
        e.step()
        e.join()
    stacked = [torch.cat([e.buffer(c, BUF_PARTIAL).clone() for e in engines]) for c in range(engines[0].E)]
    for e in engines:
        calls = {"n": 0}

        def deliver(send, recv, stream, calls=calls, e=e):
            src = stacked[calls["n"] % e.E]
            calls["n"] += 1
            torch.cuda.ExternalStream(stream, device=e.dev).synchronize()       # the exchange stream waited for the local fold
            assert torch.equal(src.view(world, -1)[e.rank], send)          # pass 2 reproduces pass 1's records
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=e.dev)):
                recv.copy_(src)
        e.set_allgather(deliver)
        e.step()
        e.join()
