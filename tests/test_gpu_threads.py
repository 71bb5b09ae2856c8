"""SURVEY §8(b) threading contract: calls on DISTINCT contexts are thread-safe (per-context streams, scratch arena,
page-locked staging, key table; thread-local error strings).  Two threads, each with its own bsx_ctx on the same GPU, drive
bsx_header_range concurrently on different inputs (ctypes releases the GIL during the calls); every result is the oracle's,
and a failing request reports ITS error on ITS thread."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import types as T

pytestmark = pytest.mark.gpu


def _call(L, ctx, J, B, V, w, r, wit=None):
    out = np.zeros(64, np.uint8)
    res = np.zeros(1, T.COMMIT_RESULT)
    inp = np.frombuffer(w.input48(r), np.uint8).copy()
    hdr = np.ascontiguousarray(w.headers[r])
    tv, rv = np.ascontiguousarray(w.validators[r]), np.ascontiguousarray(w.trusted[r])
    cid = np.frombuffer(b"celestia", np.uint8).copy()
    rc = L.bsx_header_range(ctx, C.c_uint32(J), C.c_uint32(B), _lib.p(inp), _lib.p(hdr), C.c_uint64(int(w.first_height[r])),
                            C.c_uint64(hdr.size), C.c_uint64(int(w.latest[r])), _lib.p(tv), _lib.p(rv), C.c_uint32(V), _lib.p(cid),
                            C.c_uint32(8), _lib.p(out), _lib.p(res), _lib.p(wit) if wit is not None else None)
    return rc, out.tobytes(), L.bsx_last_error().decode(errors="replace")


def test_two_contexts_two_threads():
    L = _lib.lib()
    shapes = [(8, 32, 20), (4, 16, 9)]                       # different circuits per thread: different arena / table layouts
    R, rounds = 3, 12
    ws = [synth.Workload(60 + t, R, J, B, v=V) for t, (J, B, V) in enumerate(shapes)]
    ws[1].validators[1, 2]["signature"][3] ^= 1              # thread 1, range 1: a bad signature -> its own error
    want = []
    for (J, B, V), w in zip(shapes, ws):
        want.append([oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r],
                                         w.trusted[r])[:2] for r in range(R)])
    ctxs = []
    for _ in shapes:
        h = C.c_void_p()
        assert L.bsx_init(C.c_int(0), C.byref(h)) == T.OK
        ctxs.append(h)
    errors = []

    def worker(t):
        (J, B, V), w = shapes[t], ws[t]
        try:
            for i in range(rounds):
                r = i % R
                rc, out, msg = _call(L, ctxs[t], J, B, V, w, r)
                wrc, wout = want[t][r]
                assert rc == wrc, (t, r, rc, wrc, msg)
                if rc == T.OK:
                    assert out == wout, (t, r)
                else:
                    assert "skip verification failed" in msg and "bad signatures 1" in msg, (t, msg)   # this thread's message
        except Exception as e:          # noqa: BLE001 — surfaced below
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(len(shapes))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for h in ctxs:
        L.bsx_shutdown(h)
    assert not errors, errors
    assert want[1][1][0] == T.ERR_BAD_SIGNATURE and want[0][0][0] == T.OK


def test_one_context_shared_by_two_threads():
    """the host tier keeps per-context state (arena, staging block, key table, stream2/events), and
    `_lib.context()` hands every Python thread the SAME context per device.  Host-tier calls therefore serialise on a lock
    inside the context (bsx.h): two threads hammering one context with different circuits must each get the oracle's answer —
    without the lock their arena allocations and staging bytes overlap."""
    L = _lib.lib()
    shapes = [(8, 32, 20), (4, 16, 9)]
    R, rounds = 3, 16
    ws = [synth.Workload(70 + t, R, J, B, v=V) for t, (J, B, V) in enumerate(shapes)]
    ws[0].validators[2, 1]["signature"][7] ^= 4
    want = []
    for (J, B, V), w in zip(shapes, ws):
        want.append([oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r],
                                         w.trusted[r])[:2] for r in range(R)])
    h = C.c_void_p()
    assert L.bsx_init(C.c_int(0), C.byref(h)) == T.OK
    errors = []
    go = threading.Barrier(len(shapes))

    def worker(t):
        (J, B, V), w = shapes[t], ws[t]
        try:
            go.wait()
            for i in range(rounds):
                r = i % R
                rc, out, msg = _call(L, h, J, B, V, w, r)
                wrc, wout = want[t][r]
                assert rc == wrc, (t, r, rc, wrc, msg)
                if rc == T.OK:
                    assert out == wout, (t, r)
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(len(shapes))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    L.bsx_shutdown(h)
    assert not errors, errors
    assert want[0][2][0] == T.ERR_BAD_SIGNATURE
