"""Coalescing front end (include/bsx.h `bsx_batcher_*`, `bsx_submit_*`, `bsx_wait`; csrc/batcher.hip).

The reference proves ONE range per call under a multi-thread runtime (circuits/header_range.rs:180-181) and calls the hint once per
map job — 32 `async fn hint` calls per proof (circuits/builder.rs:325-332 -> circuits/data_commitment.rs:22-44).  A `Batcher` takes
such requests from any number of threads, runs whatever arrived within a short window as ONE launch set and completes every ticket
with its own status.  ctypes releases the GIL during submit / wait, so Python threads really overlap.
"""
import ctypes as C
import threading
import weakref

import numpy as np

from . import _lib
from . import types as T


class BatcherConfig(C.Structure):
    _fields_ = [("nb_map_jobs", C.c_uint32), ("batch_size", C.c_uint32), ("v_max", C.c_uint32), ("max_requests", C.c_uint32),
                ("window_us", C.c_uint32), ("n_lanes", C.c_uint32), ("chain_id_len", C.c_uint32), ("chain_id", C.c_uint8 * 52),
                ("flags", C.c_uint32), ("key_rows", C.c_uint32), ("_reserved", C.c_uint32 * 2)]


class _KindStats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("requests", C.c_uint64), ("max_batch", C.c_uint64), ("close_wait_ns", C.c_uint64),
                ("stage_wait_ns", C.c_uint64), ("enqueue_ns", C.c_uint64), ("gpu_wait_ns", C.c_uint64), ("complete_ns", C.c_uint64)]


class BatcherStats(C.Structure):
    _fields_ = [("kind", _KindStats * 3)]


assert C.sizeof(BatcherConfig) == 96 and C.sizeof(BatcherStats) == 192
SUBMIT_INPUTS_STAY, SUBMIT_PACKED_HEADERS = 1, 2
_VIEWS = weakref.WeakSet()                   # Batcher views of context-attached batchers


def make_config(nb_map_jobs, batch_size, v_max, chain_id=b"celestia", max_requests=0, window_us=0, n_lanes=0, key_rows=0):
    cfg = BatcherConfig()
    cfg.key_rows = key_rows
    cfg.nb_map_jobs, cfg.batch_size, cfg.v_max = nb_map_jobs, batch_size, v_max
    cfg.max_requests, cfg.window_us, cfg.n_lanes = max_requests, window_us, n_lanes
    cid = bytes(chain_id)
    cfg.chain_id_len = len(cid)
    for i, x in enumerate(cid):
        cfg.chain_id[i] = x
    return cfg


class Ticket:
    """A submitted request: the ticket number and the arrays its results land in.  The arrays are ALSO held by the Batcher until the
    request has completed (a caller that drops a Ticket without waiting — an exception, fire and forget — must not free memory
    the library still writes to)."""

    def __init__(self, kind, ticket, outputs, inputs=None):
        self.kind, self.ticket, self.outputs, self.inputs = kind, ticket, outputs, inputs


def pack_headers(headers):
    """bsx_pack_headers: T.HEADER records -> one packed wire block (np.uint8 array, ~408 instead of 512 bytes per header)."""
    L = _lib.lib()
    hdr = np.ascontiguousarray(headers, T.HEADER).reshape(-1)
    L.bsx_packed_headers_bound.restype = C.c_uint64
    out = np.zeros(int(L.bsx_packed_headers_bound(C.c_uint64(hdr.size))), np.uint8)
    n = C.c_uint64(0)
    _lib.check(L.bsx_pack_headers(_lib.p(hdr), C.c_uint64(hdr.size), _lib.p(out), C.c_uint64(out.size), C.byref(n)))
    return out[:n.value].copy()


def unpack_headers(packed):
    """bsx_unpack_headers: the inverse (host code)."""
    L = _lib.lib()
    blk = np.ascontiguousarray(packed, np.uint8).reshape(-1)
    n = C.c_uint64(0)
    _lib.check(L.bsx_unpack_headers(_lib.p(blk), C.c_uint64(blk.size), None, C.c_uint64(0), C.byref(n)))
    out = np.zeros(n.value, T.HEADER)
    _lib.check(L.bsx_unpack_headers(_lib.p(blk), C.c_uint64(blk.size), _lib.p(out), C.c_uint64(out.size), C.byref(n)))
    return out


class Batcher:
    def __init__(self, nb_map_jobs, batch_size, v_max, chain_id=b"celestia", max_requests=0, window_us=0, n_lanes=0, device=0, handle=None, key_rows=0):
        self.J, self.B, self.V = nb_map_jobs, batch_size, v_max
        self.L = _lib.lib()
        self._owned = handle is None
        if handle is None:
            cfg = make_config(nb_map_jobs, batch_size, v_max, chain_id, max_requests, window_us, n_lanes, key_rows)
            h = C.c_void_p()
            _lib.check(self.L.bsx_batcher_create(_lib.context(device), C.byref(cfg), C.byref(h)))
            handle = h
        self.h = handle
        self._live = {}                       # ticket number -> (outputs, inputs that must stay): dropped once the request has completed
        self._live_mu = threading.Lock()
        if not self._owned:
            _VIEWS.add(self)                  # a view of a context's batcher: invalidated by disable_coalescing

    def _hold(self, t):
        with self._live_mu:
            self._live[t.ticket] = (t.outputs, t.inputs)
        return t

    def _release(self, number):
        with self._live_mu:
            self._live.pop(number, None)

    def _handle(self):
        if not self.h:
            raise _lib.BsxError(T.ERR_BAD_ARG, "this Batcher has been closed (or its context's coalescing was disabled)")
        return self.h

    def close(self):
        if self.h and self._owned:
            self.L.bsx_batcher_destroy(self.h)      # finishes / fails every request: nothing writes to the held arrays afterwards
        self.h = None
        with self._live_mu:
            self._live.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- CombinedSkipCircuit::define (header_range.rs:32-59)
    def submit_header_range(self, input48, headers, first_height, latest_block, target_validators, trusted_validators, packed=False, inputs_stay=False):
        """packed: `headers` is a block from pack_headers (BSX_SUBMIT_PACKED_HEADERS).  inputs_stay: `headers` is not modified until wait
        (BSX_SUBMIT_INPUTS_STAY; page-locked arrays are then uploaded from where they lie) — the Batcher holds a reference until then."""
        inp = np.frombuffer(bytes(input48), np.uint8).copy()
        if packed:
            hdr = np.ascontiguousarray(headers, np.uint8).reshape(-1)
        else:
            hdr = np.ascontiguousarray(headers, T.HEADER).reshape(-1)
        tv = np.ascontiguousarray(target_validators, T.VALIDATOR).reshape(-1)
        rv = np.ascontiguousarray(trusted_validators, T.VALIDATOR).reshape(-1)
        if tv.size != self.V or rv.size != self.V:
            raise ValueError(f"validator arrays must have {self.V} slots")
        out, res = np.zeros(64, np.uint8), np.zeros(1, T.COMMIT_RESULT)
        t = C.c_uint64(0)
        flags = (SUBMIT_PACKED_HEADERS if packed else 0) | (SUBMIT_INPUTS_STAY if inputs_stay else 0)
        _lib.check(self.L.bsx_submit_header_range_ex(self._handle(), _lib.p(inp), _lib.p(hdr), C.c_uint64(int(first_height)), C.c_uint64(hdr.size),
                                                     C.c_uint64(int(latest_block)), _lib.p(tv), _lib.p(rv), _lib.p(out), _lib.p(res), C.byref(t),
                                                     C.c_uint32(flags)))
        return self._hold(Ticket("header_range", t.value, (out, res), hdr if inputs_stay else None))

    # ---- DataCommitmentOffchainInputs::hint (data_commitment.rs:18-45 -> input.rs:149-271), MAX_LEAVES = batch_size
    def submit_data_commitment_inputs(self, headers, first_height, latest_block, start_block, end_block, want_expected=True):
        hdr = np.ascontiguousarray(headers, T.HEADER).reshape(-1)
        sh, eh = np.zeros(32, np.uint8), np.zeros(32, np.uint8)
        exp = np.zeros(32, np.uint8) if want_expected else None
        dh, lb = np.zeros(self.B, T.DH_PROOF), np.zeros(self.B, T.LB_PROOF)
        t = C.c_uint64(0)
        _lib.check(self.L.bsx_submit_data_commitment_inputs(self._handle(), _lib.p(hdr), C.c_uint64(int(first_height)), C.c_uint64(hdr.size),
                                                            C.c_uint64(int(latest_block)), C.c_uint64(int(start_block)), C.c_uint64(int(end_block)),
                                                            _lib.p(sh), _lib.p(eh), _lib.p(dh), _lib.p(lb), _lib.p(exp), C.byref(t)))
        return self._hold(Ticket("hint", t.value, (sh, eh, dh, lb, exp)))

    # ---- the map closure (builder.rs:305-336): hint + prove_subchain of one map job as ONE request
    def submit_map_job(self, range_ctx, job_index, headers, first_height, latest_block, want_proofs=True):
        """range_ctx: 1-element T.SHARED_CTX array (DataCommitmentSharedCtx).  headers[i] = header at first_height + i."""
        rg = np.ascontiguousarray(range_ctx, T.SHARED_CTX).reshape(1)
        hdr = np.ascontiguousarray(headers, T.HEADER).reshape(-1)
        sh, eh = (np.zeros(32, np.uint8), np.zeros(32, np.uint8)) if want_proofs else (None, None)
        dh, lb = (np.zeros(self.B, T.DH_PROOF), np.zeros(self.B, T.LB_PROOF)) if want_proofs else (None, None)
        rec = np.zeros(1, T.SUBCHAIN)
        t = C.c_uint64(0)
        _lib.check(self.L.bsx_submit_map_job(self._handle(), _lib.p(rg), C.c_uint32(int(job_index)), _lib.p(hdr), C.c_uint64(int(first_height)), C.c_uint64(hdr.size),
                                             C.c_uint64(int(latest_block)), _lib.p(sh), _lib.p(eh), _lib.p(dh), _lib.p(lb), _lib.p(rec), C.byref(t)))
        return self._hold(Ticket("map_job", t.value, (sh, eh, dh, lb, rec)))

    # ---- prove_subchain (builder.rs:150-271), BATCH_SIZE = batch_size
    def submit_prove_subchain(self, start_header, end_header, dh, lb, batch_start_block, batch_end_block, global_end_block, global_end_header_hash):
        dh = np.ascontiguousarray(dh, T.DH_PROOF)
        lb = np.ascontiguousarray(lb, T.LB_PROOF)
        if dh.size != self.B or lb.size != self.B:
            raise ValueError(f"proof arrays must have BATCH_SIZE = {self.B} entries")
        sh = np.frombuffer(bytes(start_header), np.uint8).copy()
        eh = np.frombuffer(bytes(end_header), np.uint8).copy()
        gh = np.frombuffer(bytes(global_end_header_hash), np.uint8).copy()
        rec = np.zeros(1, T.SUBCHAIN)
        t = C.c_uint64(0)
        _lib.check(self.L.bsx_submit_prove_subchain(self._handle(), _lib.p(sh), _lib.p(eh), _lib.p(dh), _lib.p(lb), C.c_uint64(int(batch_start_block)),
                                                    C.c_uint64(int(batch_end_block)), C.c_uint64(int(global_end_block)), _lib.p(gh), _lib.p(rec),
                                                    C.byref(t)))
        return self._hold(Ticket("subchain", t.value, (rec,)))

    def wait(self, ticket, allow=()):
        """-> (rc, results): header_range (output64 bytes, commit result); hint dict like InputDataFetcher.get_data_commitment_inputs;
        subchain record.  Raises BsxError unless the status is OK or in `allow`."""
        try:
            rc = _lib.check(self.L.bsx_wait(self._handle(), C.c_uint64(ticket.ticket)), allow=allow)
        finally:
            self._release(ticket.ticket)           # bsx_wait has returned: the request is final whatever its status
        if ticket.kind == "header_range":
            out, res = ticket.outputs
            return rc, (out.tobytes(), res[0])
        if ticket.kind == "hint":
            sh, eh, dh, lb, exp = ticket.outputs
            return rc, dict(start_header_hash=sh.tobytes(), end_header_hash=eh.tobytes(), data_hash_proofs=dh, last_block_id_proofs=lb,
                            expected_data_commitment=exp.tobytes() if exp is not None else None)
        if ticket.kind == "map_job":
            sh, eh, dh, lb, rec = ticket.outputs
            return rc, dict(start_header_hash=sh.tobytes() if sh is not None else None, end_header_hash=eh.tobytes() if eh is not None else None,
                            data_hash_proofs=dh, last_block_id_proofs=lb, record=rec[0])
        return rc, ticket.outputs[0][0]

    def done(self, ticket):
        d = C.c_int(0)
        _lib.check(self.L.bsx_poll(self._handle(), C.c_uint64(ticket.ticket), C.byref(d)))
        if d.value:
            self._release(ticket.ticket)
        return bool(d.value)

    def cork(self, on=True):
        """While corked, open batches close only when full (announce a burst, submit it, uncork)."""
        _lib.check(self.L.bsx_batcher_cork(self._handle(), C.c_int(1 if on else 0)))

    def stats(self):
        s = BatcherStats()
        _lib.check(self.L.bsx_batcher_get_stats(self._handle(), C.byref(s)))
        names = ("header_range", "data_commitment_inputs", "prove_subchain")
        def row(k):
            nb = max(1, int(k.batches))
            return {"batches": int(k.batches), "requests": int(k.requests), "max_batch": int(k.max_batch), "close_wait_us": k.close_wait_ns / 1e3 / nb,
                    "stage_wait_us": k.stage_wait_ns / 1e3 / nb, "enqueue_us": k.enqueue_ns / 1e3 / nb, "gpu_wait_us": k.gpu_wait_ns / 1e3 / nb,
                    "complete_us": k.complete_ns / 1e3 / nb}
        return {n: row(s.kind[i]) for i, n in enumerate(names)}


def enable_coalescing(nb_map_jobs, batch_size, v_max, chain_id=b"celestia", max_requests=0, window_us=0, n_lanes=0, device=0):
    """bsx_enable_coalescing on the process's context of `device`: the synchronous builder calls (CombinedSkipCircuit.prove without a
    witness, InputDataFetcher.get_data_commitment_inputs with MAX_LEAVES = batch_size, DataCommitmentBuilder.prove_subchain without a
    witness) made from any number of threads coalesce.  Returns a Batcher view of the attached batcher (stats)."""
    cfg = make_config(nb_map_jobs, batch_size, v_max, chain_id, max_requests, window_us, n_lanes)
    L = _lib.lib()
    _lib.check(L.bsx_enable_coalescing(_lib.context(device), C.byref(cfg)))
    L.bsx_context_batcher.restype = C.c_void_p
    return Batcher(nb_map_jobs, batch_size, v_max, handle=C.c_void_p(L.bsx_context_batcher(_lib.context(device))))


def disable_coalescing(device=0):
    _lib.check(_lib.lib().bsx_enable_coalescing(_lib.context(device), None))
    for v in list(_VIEWS):                    # the attached batcher is gone: views of it must not hand out a dangling handle
        v.h = None
        with v._live_mu:
            v._live.clear()
