"""Mode S (BASELINE configs #4/#5: "N headers x V validators"): a V-validator commit on EVERY header of a range, i.e. N
back-to-back next_header / skip verifications (circuits/next_header.rs:25-47 per header; batches are independent,
circuits/builder.rs:305-336), sharded with the headers across GPUs.

CommitShard is a thin wrapper over ONE C-ABI call per step — `bsx_dev_verify_commits` (include/bsx.h): SHA-512 challenges,
fixed-key Ed25519, tallies + validator-set hashes and the 128-byte fold of the slice — plus the one collective of the mode:
an all-gather of the ranks' folds (torch.distributed: RCCL on GPUs, gloo in the CPU tests).  PyTorch owns the device buffers
and the collective; no arithmetic happens here.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import types as T


def commit_slice(n_commits, rank, world):
    """Commits [first, first + count) of the range that `rank` verifies: contiguous, equal slices (2048 / N headers each)."""
    assert n_commits % world == 0, "world size must divide the number of headers"
    count = n_commits // world
    return rank * count, count


def all_gather_folds(fold_bytes, world):
    """THE collective of mode S: every rank's 128-byte bsx_commit_fold -> uint8 [world][128] on every rank."""
    import torch
    import torch.distributed as dist
    flat = fold_bytes[:128].contiguous()
    if world == 1:
        return flat.clone().view(1, 128)
    if flat.is_cuda and dist.get_backend() == "gloo":          # tests: ranks sharing one GPU
        g = torch.empty(world * 128, dtype=torch.uint8)
        dist.all_gather_into_tensor(g, flat.cpu())
        return g.view(world, 128).to(flat.device)
    out = torch.empty(world * 128, dtype=torch.uint8, device=flat.device)
    dist.all_gather_into_tensor(out, flat)
    return out.view(world, 128)


def range_verdict(folds):
    """folds: COMMIT_FOLD[world] -> dict: did every commit of the whole range verify, first failing global index, checksum."""
    import hashlib
    f = np.asarray(folds, T.COMMIT_FOLD)
    ff = f["first_failing"][f["first_failing"] != 0xffffffff]
    return {"commits": int(f["n_commits"].sum()), "ok": int(f["n_ok"].sum()), "signatures_ok": int(f["n_signatures_ok"].sum()),
            "all_ok": bool(f["n_ok"].sum() == f["n_commits"].sum()), "first_failing": int(ff.min()) if ff.size else None,
            "root_of_roots": hashlib.sha256(b"".join(bytes(x["root"]) for x in f)).hexdigest()}


class CommitShard:
    def __init__(self, n_commits_total, v_max, rank=0, world=1, device=None, with_witness=False, expand=False, n_sets=1, tally_beside=True, wide_tables=None):
        """with_witness: every step also leaves the commits' COMMIT units (include/bsx_layout.h: the Goldilocks witness of the
        per-validator loop, BASELINE config #5) in compact form in `self.compact`; expand: and expands them into `self.witness`
        (u64 [n][commit_layout(V).n_elements]: 15 MB per commit at V = 512) on the same stream.
        n_sets = K > 1: K sets of output buffers (verdicts, results, fold, scratch, units) and K streams; step i runs on set i mod K,
        so that step i + 1 starts while step i's stages drain (a stage is a few resident rounds of long waves: alone, its ramp-up,
        its last partial round and the SIMDs with one wave less than their neighbours idle).  The inputs and the key tables are shared,
        read only.  `ok` / `res` / `fold` / `compact` / `witness` name the set of the LAST step.
        wide_tables: BSX_COMMITS_KEYTABLE_WIDE — 16-bit digits in the key tables (64 MB per key; 16 + 16 instead of 22 + 16 additions per
        signature).  Default: on (6.4 GB of tables at V = 100, 34 GB at V = 512 — of 288 GB; a shard's validator set is resident:
        2048 x 512 verification 2.22 -> 1.94 ms).
        tally_beside: BSX_COMMITS_TALLY_BESIDE — the validator-set trees run beside the signature check on the context's side stream
        (one thread drives a shard, as the flag requires)."""
        import torch
        self.N, self.V, self.rank, self.world = n_commits_total, v_max, rank, world
        self.first, self.n = commit_slice(n_commits_total, rank, world)
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.ctx = _lib.context(self.dev.index if self.dev.index is not None else 0)
        self.L = _lib.lib()
        n, V, d = self.n, v_max, self.dev
        z = lambda nbytes: torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=d)
        self.K = int(n_sets)
        self.tally_beside = bool(tally_beside)
        assert self.K >= 1
        self.vals = z(n * V * 256)
        self.hh = z(n * 32)
        self.kt_bits = 16 if (wide_tables if wide_tables is not None else True) else 12
        self.keytable = z(int(self.L.bsx_ed25519_keytable_bytes_w(C.c_uint32(V), C.c_uint32(self.kt_bits))))
        self.lay = T.commit_layout(V)
        self._sets = []
        for _ in range(self.K):
            st = {"ok": z(n * V), "res": z(n * 96), "fold": z(128),
                  "scratch": z(int(self.L.bsx_dev_verify_commits_scratch_bytes(C.c_uint32(n), C.c_uint32(V)))),
                  "compact": z(n * int(self.lay["compact_stride"])) if (with_witness or expand) else None, "witness": None, "wbuf": None,
                  "stream": torch.cuda.Stream(d) if self.K > 1 else None}
            if expand:
                st["wbuf"] = _lib.DeviceBuffer(n * int(self.lay["n_elements"]) + 2, self.dev.index if self.dev.index is not None else 0)
                st["witness"] = st["wbuf"].tensor()
            self._sets.append(st)
        self.steps = 0
        self._use(0)

    def _use(self, k):
        s = self._sets[k]
        self.cur = k
        self.ok, self.res, self.fold, self.scratch, self.compact, self.witness = s["ok"], s["res"], s["fold"], s["scratch"], s["compact"], s["witness"]

    def upload(self, validators, header_hashes):
        """validators [N, V] VALIDATOR and header_hashes [N, 32] of the WHOLE range; this rank keeps its slice."""
        import torch
        v = np.ascontiguousarray(validators, T.VALIDATOR).reshape(self.N, self.V)[self.first:self.first + self.n]
        h = np.ascontiguousarray(header_hashes, np.uint8).reshape(self.N, 32)[self.first:self.first + self.n]
        self.vals[:v.nbytes].copy_(torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1)))
        self.hh[:h.nbytes].copy_(torch.from_numpy(np.ascontiguousarray(h).reshape(-1)))
        torch.cuda.synchronize(self.dev)
        # validator sets change here and nowhere else: the fixed-key table is checked / rebuilt now (bsx_dev_ed25519_keytable), and
        # the keys are compared on the host, so that a step launches neither the key compare nor — when every active slot carries
        # the first commit's key of its index — the generic-kernel pass (bsx.h BSX_COMMITS_*)
        st = torch.cuda.current_stream(self.dev)
        _lib.check(self.L.bsx_dev_ed25519_keytable_w(self.ctx, C.c_void_p(st.cuda_stream), _lib.dp(self.vals), C.c_uint32(self.V), _lib.dp(self.keytable),
                                                     C.c_uint32(self.kt_bits)))
        torch.cuda.synchronize(self.dev)
        active = (v["enabled"] != 0) & (v["is_signed"] != 0)
        uniform = bool((~active | (v["pubkey"] == v["pubkey"][0][None]).all(axis=2)).all())
        self.flags = 1 | (2 if uniform else 0) | (4 if self.tally_beside else 0) | (8 if self.kt_bits == 16 else 0)

    def step(self, stream=None):
        """One bsx_dev_verify_commits over this rank's slice [+ the units' expansion], on `stream` (default: the current stream; with
        n_sets > 1 the set's own stream).  Returns the index of the buffer set it runs on."""
        import torch
        k = self.steps % self.K
        self.steps += 1
        self._use(k)
        st = stream if stream is not None else (self._sets[k]["stream"] or torch.cuda.current_stream(self.dev))
        dp = _lib.dp
        _lib.check(self.L.bsx_dev_verify_commits(self.ctx, C.c_void_p(st.cuda_stream), dp(self.vals), C.c_uint32(self.n), C.c_uint32(self.V),
                                                 dp(self.hh), C.c_uint32(self.first), dp(self.keytable), dp(self.scratch), dp(self.ok),
                                                 dp(self.res), dp(self.fold), dp(self.compact), C.c_uint32(getattr(self, "flags", 0))))
        if self.witness is not None:
            lay = np.ascontiguousarray(self.lay).reshape(1)
            _lib.check(self.L.bsx_dev_expand_witness(self.ctx, C.c_void_p(st.cuda_stream), _lib.p(lay), C.c_uint32(self.n), dp(self.compact),
                                                     dp(self.witness)))
        return k

    def witness_of(self, commits):
        """Expanded COMMIT units of the given local commit indices -> u64 [len(commits), n_elements] (host)."""
        import torch
        torch.cuda.synchronize(self.dev)
        nel = int(self.lay["n_elements"])
        return np.stack([self.witness[c * nel:(c + 1) * nel].cpu().numpy().view(np.uint64) for c in commits])

    def compact_of(self, commits):
        import torch
        torch.cuda.synchronize(self.dev)
        cs = int(self.lay["compact_stride"])
        return np.stack([self.compact[c * cs:(c + 1) * cs].cpu().numpy() for c in commits])

    def gather(self, k=None):
        """All ranks' folds -> COMMIT_FOLD[world] (host).  k: the buffer set whose step to gather (default: the last step's); with
        n_sets > 1 the collective and the copy are ordered on that set's stream, the other sets' steps keep running."""
        import torch
        s = self._sets[self.cur if k is None else k]
        if s["stream"] is None:
            return all_gather_folds(s["fold"], self.world).cpu().numpy().reshape(-1).view(T.COMMIT_FOLD).copy()
        with torch.cuda.stream(s["stream"]):
            g = all_gather_folds(s["fold"], self.world)
            h = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
            h.copy_(g, non_blocking=True)
        s["stream"].synchronize()
        return h.numpy().reshape(-1).view(T.COMMIT_FOLD).copy()

    def download(self):
        import torch
        torch.cuda.synchronize(self.dev)
        return (self.ok[:self.n * self.V].cpu().numpy().reshape(self.n, self.V), self.res[:self.n * 96].cpu().numpy().view(T.COMMIT_RESULT).copy(),
                self.fold.cpu().numpy().view(T.COMMIT_FOLD)[0].copy())
