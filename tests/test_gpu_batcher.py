"""The coalescing front end (include/bsx.h bsx_batcher_* / bsx_submit_* / bsx_wait; csrc/batcher.hip) against the oracle.

The reference's call shape — one range per `prove` call under a multi-thread runtime (/root/reference/circuits/header_range.rs:180-181),
one `async fn hint` per map job (/root/reference/circuits/builder.rs:325-332 -> /root/reference/circuits/data_commitment.rs:22-44) — is
served by coalescing whatever requests arrive together into ONE launch set.  What must hold, whatever the batching happened to be:
every request gets the oracle's output, commit result and STATUS (a tampered / malformed request never fails its batch-mates), from
any number of threads, through the explicit submit / wait calls and through the unchanged synchronous calls on a context with
coalescing enabled."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import batcher as BT
from blobstreamx_amd import types as T
from blobstreamx_amd.builder import CombinedSkipCircuit, DataCommitmentBuilder, InputDataFetcher

pytestmark = pytest.mark.gpu


def _tamper(w, r, kind):
    """In-place corruption of range r of a synth.Workload; every kind fails differently (and `ok` not at all)."""
    n = w.n_blocks
    if kind == "sig":                   # a signed validator's signature -> BSX_ERR_BAD_SIGNATURE
        w.validators[r, 1 + r % 3]["signature"][5] ^= 0x10
    elif kind == "link":                # a header in the middle: its hash changes -> the chain link breaks (A3 / A4 / A8 ...)
        w.headers[r, n // 2]["proposer"][5] ^= 1
    elif kind == "data_hash":           # the data hash of a header: header hash changes too
        w.headers[r, max(1, n // 3)]["hash"][1][7] ^= 0x80
    elif kind == "power":               # most of the voting power did not sign -> BSX_ERR_VOTING_POWER
        for v in range(w.validators.shape[1] // 2 + 1):
            w.validators[r, v]["is_signed"] = 0
    elif kind == "trusted_hash":        # public input's trusted header hash is wrong
        w.ranges[r]["start_header_hash"][3] ^= 2
    elif kind == "bad_header":          # malformed packed header: a field length beyond its capacity -> BSX_ERR_BAD_HEADER
        w.headers[r, 2]["len"][3] = 60
    elif kind == "chain":               # the target header belongs to another chain (field 1)
        w.headers[r, n]["chain_id"][3] ^= 1
    else:
        assert kind == "ok"


KINDS = ["ok", "sig", "ok", "link", "ok", "data_hash", "power", "ok", "trusted_hash", "bad_header", "ok", "chain"]


def _oracle_all(w, J, B):
    return [oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])[:3]
            for r in range(w.R)]


def test_64_mixed_requests_from_16_threads_vs_oracle():
    """64 good / tampered requests from 16 threads through submit + wait: every output, commit result and status is the
    oracle's, and the requests did coalesce (fewer launch sets than requests)."""
    J, B, V, R, NT = 8, 32, 20, 64, 16
    w = synth.Workload(81, R, J, B, v=V, n_blocks=J * B - 5)
    kinds = [KINDS[r % len(KINDS)] for r in range(R)]
    for r, k in enumerate(kinds):
        _tamper(w, r, k)
    want = _oracle_all(w, J, B)
    assert len({rc for rc, _, _ in want}) >= 5, "the tampering must produce a spread of statuses"
    bt = BT.Batcher(J, B, V, max_requests=16)
    got, errors = [None] * R, []
    go = threading.Barrier(NT)

    def worker(t):
        try:
            go.wait()
            mine = list(range(t, R, NT))
            for rep in range(3):                 # every request three times: batches of different composition
                tickets = [(r, bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]))
                           for r in mine[rep % 2::1]]
                for r, tk in tickets:
                    rc, (out, res) = bt.wait(tk, allow=tuple(range(1, 10)))
                    got[r] = (rc, out, res.copy(), _lib.last_error() if rc else "")
        except Exception as e:      # noqa: BLE001 — surfaced below
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for r in range(R):
        wrc, wout, wres = want[r]
        rc, out, res, msg = got[r]
        assert rc == wrc, (r, kinds[r], rc, wrc, msg)
        if rc in (T.OK, T.ERR_ASSERT, T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER):
            assert out == wout, (r, kinds[r])
            assert res.tobytes() == wres.tobytes(), (r, kinds[r], res, wres)
        if rc == T.ERR_BAD_SIGNATURE:
            assert "bad signatures 1" in msg, msg            # THIS request's message on the waiting thread
    st = bt.stats()["header_range"]
    assert st["requests"] >= 3 * R - NT and st["batches"] < st["requests"], st       # coalesced
    assert st["max_batch"] > 1, st
    bt.close()


def test_one_submitter_fills_batches_and_statuses_stay_apart():
    """40 tickets from ONE thread before the first wait: they fill batches of 16 (max_requests); the malformed and the tampered requests
    sit in the same launch sets as good ones and only they fail."""
    J, B, V, R = 4, 16, 7, 40
    w = synth.Workload(82, R, J, B, v=V)
    kinds = ["bad_header" if r % 16 == 3 else "sig" if r % 16 == 9 else "link" if r % 16 == 12 else "ok" for r in range(R)]
    for r, k in enumerate(kinds):
        _tamper(w, r, k)
    want = _oracle_all(w, J, B)
    bt = BT.Batcher(J, B, V, max_requests=16)
    bt.cork()                                    # a burst is coming: batches close when full ...
    tickets = [bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]) for r in range(R)]
    bt.cork(False)                               # ... and the last, partial one now
    for r in reversed(range(R)):                 # waited for in another order than submitted
        rc, (out, res) = bt.wait(tickets[r], allow=tuple(range(1, 10)))
        assert rc == want[r][0], (r, kinds[r], rc, want[r][0], _lib.last_error())
        if rc != T.ERR_BAD_HEADER:
            assert out == want[r][1] and res.tobytes() == want[r][2].tobytes(), (r, kinds[r])
        assert bt.done(tickets[r])
    assert [want[r][0] for r in (3, 9, 12, 0)] == [T.ERR_BAD_HEADER, T.ERR_BAD_SIGNATURE, T.ERR_ASSERT, T.OK]
    st = bt.stats()["header_range"]
    assert st["max_batch"] == 16 and st["batches"] == 3, st
    # a ticket can be waited for again; an unknown one is refused
    assert bt.L.bsx_wait(bt.h, C.c_uint64(tickets[0].ticket)) == T.OK
    assert bt.L.bsx_wait(bt.h, C.c_uint64(10 ** 9)) == T.ERR_BAD_ARG
    bt.close()


def test_submit_refuses_what_the_arguments_alone_show():
    J, B, V = 4, 16, 5
    w = synth.Workload(83, 1, J, B, v=V)
    bt = BT.Batcher(J, B, V)
    S = int(w.first_height[0])
    with pytest.raises(_lib.BsxError) as e:      # target <= trusted
        bt.submit_header_range(S.to_bytes(8, "big") + bytes(32) + S.to_bytes(8, "big"), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0])
    assert e.value.status == T.ERR_RANGE_TOO_LONG
    with pytest.raises(_lib.BsxError) as e:      # the target header is not among the supplied ones
        bt.submit_header_range(w.input48(0), w.headers[0][:10], S, int(w.latest[0]), w.validators[0], w.trusted[0])
    assert e.value.status == T.ERR_BAD_ARG
    with pytest.raises(_lib.BsxError) as e:      # hint: end - start > MAX_LEAVES (input.rs:154)
        bt.submit_data_commitment_inputs(w.headers[0], S, int(w.latest[0]), S, S + B + 1)
    assert e.value.status == T.ERR_RANGE_TOO_LONG
    with pytest.raises(_lib.BsxError) as e:      # hint: the headers do not reach min(end, latest - 2)
        bt.submit_data_commitment_inputs(w.headers[0][:5], S, int(w.latest[0]), S, S + B)
    assert e.value.status == T.ERR_BAD_ARG
    # nothing was issued: the batcher is still usable
    rc, (out, _) = bt.wait(bt.submit_header_range(w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0]))
    assert rc == T.OK and out == oracle.header_range(J, B, w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0])[1]
    bt.close()


@pytest.mark.parametrize("J,B", [(32, 64), (8, 16)])
def test_the_map_job_hints_of_a_proof_from_one_thread_each_vs_oracle(J, B):
    """The 32 `async fn hint` calls of one header_range_2048 (builder.rs:325-332), one thread each, followed by prove_subchain
    (builder.rs:335) on what the hint returned: every hint output (start / end header, both proof arrays, expected commitment) and every
    map-job record is the oracle's — including the clamped, zero-padded and dummy-header batches behind a short range, and a tampered one."""
    V = 4
    n_blocks = J * B - 3 * B - 7                  # the last batches: partly enabled, then disabled (zero proofs, dummy headers)
    w = synth.Workload(84, 1, J, B, v=V, n_blocks=n_blocks)
    w.latest[0] = int(w.first_height[0]) + n_blocks + 1          # chain head right behind the target: latest - 2 clamps the hint
    w.headers[0, 2 * B + 3]["hash"][1][0] ^= 1                   # job 2: a data hash that is not the one its successor committed to
    S, latest = int(w.first_height[0]), int(w.latest[0])
    E = S + n_blocks
    hh = oracle.header_hash_only(w.headers[0])
    bt = BT.Batcher(J, B, V)
    hint_out, rec_out, errors = [None] * J, [None] * J, []
    go = threading.Barrier(J)

    def worker(j):
        try:
            bs, be = S + j * B, S + (j + 1) * B
            hdrs = w.headers[0][j * B:(j + 1) * B + 1]
            go.wait()
            rc, h = bt.wait(bt.submit_data_commitment_inputs(hdrs, bs, latest, bs, be))
            hint_out[j] = h
            tk = bt.submit_prove_subchain(h["start_header_hash"], h["end_header_hash"], h["data_hash_proofs"], h["last_block_id_proofs"], bs, be, E,
                                          hh[n_blocks].tobytes())
            rc, rec = bt.wait(tk, allow=(T.ERR_ASSERT,))
            rec_out[j] = (rc, rec.copy())
        except Exception as e:      # noqa: BLE001
            errors.append((j, repr(e)))

    th = [threading.Thread(target=worker, args=(j,)) for j in range(J)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    n_fail = 0
    for j in range(J):
        bs, be = S + j * B, S + (j + 1) * B
        orc, oh = oracle.data_commitment_inputs(w.headers[0][j * B:(j + 1) * B + 1], bs, latest, bs, be, B)
        assert orc == T.OK
        h = hint_out[j]
        assert h["start_header_hash"] == oh["start_header"] and h["end_header_hash"] == oh["end_header"], j
        assert h["data_hash_proofs"].tobytes() == oh["data_hash_proofs"].tobytes(), j
        assert h["last_block_id_proofs"].tobytes() == oh["last_block_id_proofs"].tobytes(), j
        assert h["expected_data_commitment"] == oh["expected_data_commitment"], j
        wrc, wrec, _ = oracle.prove_subchain(B, oh["start_header"], oh["end_header"], oh["data_hash_proofs"], oh["last_block_id_proofs"], bs, be, E,
                                             hh[n_blocks].tobytes())
        rc, rec = rec_out[j]
        assert rc == wrc and rec.tobytes() == wrec.tobytes(), (j, rc, wrc, rec, wrec)
        n_fail += rc != T.OK
    assert n_fail >= 1                           # the tampered job failed, the others did not
    st = bt.stats()
    assert st["data_commitment_inputs"]["requests"] == J and st["prove_subchain"]["requests"] == J
    assert st["data_commitment_inputs"]["batches"] < J, st
    bt.close()


def test_synchronous_calls_on_a_context_with_coalescing_enabled():
    """bsx_enable_coalescing: the UNCHANGED builder calls, from 8 threads sharing the process's context, coalesce; a call with a witness
    and a call of another circuit shape take the serial path on the same context, beside them."""
    J, B, V, R, NT = 8, 32, 10, 16, 8
    w = synth.Workload(85, R, J, B, v=V)
    _tamper(w, 3, "sig")
    _tamper(w, 6, "link")
    want = _oracle_all(w, J, B)
    view = BT.enable_coalescing(J, B, V)
    try:
        circ = [CombinedSkipCircuit(V, J, B) for _ in range(NT)]
        errors, got = [], [None] * R
        go = threading.Barrier(NT)

        def worker(t):
            try:
                go.wait()
                for rep in range(2):
                    for r in range(t, R, NT):
                        f = InputDataFetcher(w.headers[r], int(w.first_height[r]), int(w.latest[r]))
                        out, res, _ = circ[t].prove(w.input48(r), f, w.validators[r], w.trusted[r], allow=tuple(range(1, 10)))
                        got[r] = (circ[t].last_rc, out, res.copy())
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
        for x in th:
            x.start()
        # meanwhile, on the main thread: a witness request (serial path) and the hint + prove_subchain of one map job (coalesced kinds)
        f0 = InputDataFetcher(w.headers[0], int(w.first_height[0]), int(w.latest[0]))
        out_w, _, wit = CombinedSkipCircuit(V, J, B).prove(w.input48(0), f0, w.validators[0], w.trusted[0], want_witness=True)
        S = int(w.first_height[0])
        h = f0.get_data_commitment_inputs(S + B, S + 2 * B, B)
        rec, _ = DataCommitmentBuilder().prove_subchain(h, S + B, S + 2 * B, S + J * B, w.hashes[0, J * B].tobytes())
        # another shape on the same context: serial
        w2 = synth.Workload(86, 1, 4, 16, v=V)
        out2, _, _ = CombinedSkipCircuit(V, 4, 16).prove(w2.input48(0), InputDataFetcher(w2.headers[0], int(w2.first_height[0]), int(w2.latest[0])),
                                                         w2.validators[0], w2.trusted[0])
        for x in th:
            x.join()
        assert not errors, errors
        for r in range(R):
            assert got[r][0] == want[r][0] and got[r][1] == want[r][1] and got[r][2].tobytes() == want[r][2].tobytes(), r
        assert [want[r][0] for r in (3, 6)] == [T.ERR_BAD_SIGNATURE, T.ERR_ASSERT]
        rc, ref_out, _, cw = oracle.header_range(J, B, w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0], want_witness=True)
        assert out_w == ref_out and (wit == oracle.expand_range_witness(J, B, cw, v_max=V)).all()
        orc, oh = oracle.data_commitment_inputs(w.headers[0], S, int(w.latest[0]), S + B, S + 2 * B, B)
        assert h["data_hash_proofs"].tobytes() == oh["data_hash_proofs"].tobytes() and h["expected_data_commitment"] == oh["expected_data_commitment"]
        wrc, wrec, _ = oracle.prove_subchain(B, oh["start_header"], oh["end_header"], oh["data_hash_proofs"], oh["last_block_id_proofs"], S + B, S + 2 * B,
                                             S + J * B, w.hashes[0, J * B].tobytes())
        assert wrc == T.OK and rec.tobytes() == wrec.tobytes()
        assert out2 == oracle.header_range(4, 16, w2.input48(0), w2.headers[0], int(w2.first_height[0]), int(w2.latest[0]), w2.validators[0], w2.trusted[0])[1]
        st = view.stats()
            # a caller that finds the batcher idle and the context's serial path free runs there (a lone caller pays no hops): most of the 2 R
        # calls coalesced, a few took the serial path — every answer was checked above either way
        assert R <= st["header_range"]["requests"] <= 2 * R and st["data_commitment_inputs"]["requests"] == 1 and st["prove_subchain"]["requests"] == 1, st
    finally:
        BT.disable_coalescing()


def test_validator_set_changes_between_and_inside_batches():
    """The lanes' fixed-key tables follow the validator set of the batch's first request; requests signed by ANOTHER set in the same
    batch (slots whose key differs) are verified by the generic kernel: same verdicts as the oracle either way."""
    J, B, R = 4, 16, 12
    wa = synth.Workload(87, R, J, B, v=9)
    wb = synth.Workload(88, R, J, B, v=9)       # other seed: other keys
    wb.validators[5, 2]["signature"][0] ^= 1
    bt = BT.Batcher(J, B, 9, max_requests=8)
    order = [(wa, r) if (r // 2) % 2 == 0 else (wb, r) for r in range(R)]      # a a b b a a b b ...
    bt.cork()
    tickets = [bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]) for w, r in order]
    bt.cork(False)
    for (w, r), tk in zip(order, tickets):
        rc, (out, res) = bt.wait(tk, allow=(T.ERR_BAD_SIGNATURE,))
        wrc, wout, wres = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])[:3]
        assert (rc, out) == (wrc, wout) and res.tobytes() == wres.tobytes(), (r, rc, wrc)
    assert bt.stats()["header_range"]["max_batch"] == 8
    bt.close()


@pytest.mark.parametrize("coalesce", [True, False])
def test_map_job_one_call_vs_oracle(coalesce):
    """bsx_map_job = the map closure of prove_data_commitment for one map job (builder.rs:305-336: hint, then prove_subchain) as ONE
    call: through a batcher (threads, coalesced, path digests from the header trees) and on a plain context (the two host-tier calls in
    turn) — proofs and record are the oracle's for every job of a short, tampered range, including the batches the chain head does not
    reach."""
    J, B, V = 16, 32, 4
    n_blocks = J * B - 2 * B - 9
    w = synth.Workload(89, 1, J, B, v=V, n_blocks=n_blocks)
    w.latest[0] = int(w.first_height[0]) + n_blocks + 1
    w.headers[0, 3 * B + 5]["hash"][1][9] ^= 4
    S, latest, E = int(w.first_height[0]), int(w.latest[0]), int(w.first_height[0]) + n_blocks
    hh = oracle.header_hash_only(w.headers[0])
    rg = np.zeros(1, T.SHARED_CTX)
    rg["start_block"], rg["end_block"] = S, E
    rg["start_header_hash"][0] = hh[0]
    rg["end_header_hash"][0] = hh[n_blocks]
    L = _lib.lib()
    got = [None] * J
    if coalesce:
        bt = BT.Batcher(J, B, V)
        errors = []

        def worker(j):
            try:
                rc, r = bt.wait(bt.submit_map_job(rg, j, w.headers[0][j * B:(j + 1) * B + 1], S + j * B, latest), allow=(T.ERR_ASSERT,))
                got[j] = (rc, r["start_header_hash"], r["end_header_hash"], r["data_hash_proofs"], r["last_block_id_proofs"], r["record"].copy())
            except Exception as e:      # noqa: BLE001
                errors.append((j, repr(e)))
        th = [threading.Thread(target=worker, args=(j,)) for j in range(J)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors
        assert bt.stats()["data_commitment_inputs"]["requests"] == J
        bt.close()
    else:
        h = C.c_void_p()
        assert L.bsx_init(C.c_int(0), C.byref(h)) == T.OK
        for j in range(J):
            hdrs = np.ascontiguousarray(w.headers[0][j * B:(j + 1) * B + 1])
            sh, eh, dh, lb, rec = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(B, T.DH_PROOF), np.zeros(B, T.LB_PROOF), np.zeros(1, T.SUBCHAIN)
            rc = L.bsx_map_job(h, C.c_uint32(J), C.c_uint32(B), _lib.p(rg), C.c_uint32(j), _lib.p(hdrs), C.c_uint64(S + j * B), C.c_uint64(hdrs.size),
                               C.c_uint64(latest), _lib.p(sh), _lib.p(eh), _lib.p(dh), _lib.p(lb), _lib.p(rec))
            assert rc in (T.OK, T.ERR_ASSERT), (j, rc, _lib.last_error())
            got[j] = (rc, sh.tobytes(), eh.tobytes(), dh, lb, rec[0].copy())
        L.bsx_shutdown(h)
    n_fail = 0
    for j in range(J):
        bs, be = S + j * B, S + (j + 1) * B
        orc, oh = oracle.data_commitment_inputs(w.headers[0][j * B:(j + 1) * B + 1], bs, latest, bs, be, B)
        assert orc == T.OK
        wrc, wrec, _ = oracle.prove_subchain(B, oh["start_header"], oh["end_header"], oh["data_hash_proofs"], oh["last_block_id_proofs"], bs, be, E,
                                             hh[n_blocks].tobytes())
        rc, sh, eh, dh, lb, rec = got[j]
        assert (sh, eh) == (oh["start_header"], oh["end_header"]), j
        assert dh.tobytes() == oh["data_hash_proofs"].tobytes() and lb.tobytes() == oh["last_block_id_proofs"].tobytes(), j
        assert rc == wrc and rec.tobytes() == wrec.tobytes(), (j, rc, wrc, rec, wrec)
        n_fail += rc != T.OK
    assert n_fail >= 1


def test_rotating_validator_sets_through_one_batcher():
    """Requests signed by validator sets that drift from request to request (synth rotate_permille) in the same launch sets: the lanes'
    tables are keyed by public key (csrc/keycache.h), results are the oracle's; with a table too small for the keys of one launch set
    (key_rows = v_max) the overflow goes to the generic kernel — same results."""
    J, B, V, R = 4, 16, 12, 24
    w = synth.Workload(91, R, J, B, v=V, rotate_permille=250)
    w.validators[R - 2, 0]["signature"][1] ^= 1
    want = _oracle_all(w, J, B)
    for key_rows in (0, V):
        bt = BT.Batcher(J, B, V, max_requests=8, key_rows=key_rows)
        for rep in range(2):
            bt.cork()
            tickets = [bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]) for r in range(R)]
            bt.cork(False)
            for r, tk in enumerate(tickets):
                rc, (out, res) = bt.wait(tk, allow=(T.ERR_BAD_SIGNATURE,))
                assert (rc, out) == want[r][:2] and res.tobytes() == want[r][2].tobytes(), (key_rows, rep, r, rc, want[r][0])
        assert want[R - 2][0] == T.ERR_BAD_SIGNATURE
        bt.close()


# ---------------------------------------------------------------------------------------------------------------- round 6
def test_packed_wire_headers_on_the_fixtures_vs_node_reported_values(mocha, golden):
    """VERDICT r5 #2c: the fixture chain 10000 -> 10004 (/root/reference/circuits/fixtures/mocha-4, committed as tests/golden) with its
    headers PACKED (bsx_pack_headers: ~400 instead of 512 bytes per header over PCIe, laid out as records by k_unpack_headers): the public
    output is (the node's hash of block 10004, the node-reported data commitment 10000-10004) — the same bytes the 512-byte records give
    and the oracle computes."""
    hdr = mocha["headers"]
    pk = BT.pack_headers(hdr)
    assert pk.size < hdr.nbytes * 0.85 and (BT.unpack_headers(pk).view(np.uint8) == np.ascontiguousarray(hdr).view(np.uint8)).all()
    inp = bytes.fromhex(golden["kats"]["header_range_input_10000_10004"])
    tv = mocha["commits"][4]
    rv = mocha["commits"][0].copy()
    rv["is_signed"] = 0
    rc, want_out, want_res = oracle.header_range(2, 2, inp, hdr, 10000, 10006, tv, rv, chain_id=b"mocha-4")[:3]
    assert rc == T.OK and want_out[:32] == mocha["hashes"][4] and want_out[32:].hex() == golden["data_commitments"]["10000-10004"]
    bt = BT.Batcher(2, 2, 4, chain_id=b"mocha-4")
    for packed in (True, False, True):
        t = bt.submit_header_range(inp, pk if packed else hdr, 10000, 10006, tv, rv, packed=packed)
        rc, (out, res) = bt.wait(t)
        assert rc == T.OK and out == want_out and res.tobytes() == want_res.tobytes(), packed
    # a block that starts BEFORE the trusted header (first_height < trusted): the request takes its part of the block
    inp2 = (10002).to_bytes(8, "big") + mocha["hashes"][2] + (10004).to_bytes(8, "big")
    rv2 = mocha["commits"][2].copy()
    rv2["is_signed"] = 0
    rc2, want2, _ = oracle.header_range(2, 2, inp2, hdr, 10000, 10006, tv, rv2, chain_id=b"mocha-4")[:3]
    t = bt.submit_header_range(inp2, pk, 10000, 10006, tv, rv2, packed=True)
    rc, (out, _) = bt.wait(t)
    assert rc == rc2 == T.OK and out == want2 and out[32:].hex() == golden["data_commitments"]["10002-10004"]
    bt.close()


def test_packed_and_record_requests_share_launch_sets_vs_oracle():
    """Packed and 512-byte requests, good and tampered, of different lengths, interleaved in the SAME batches (slots change form from one
    batch to the next, shorter requests follow longer ones in a slot): every output / commit result / status is the oracle's."""
    J, B, V, R = 4, 16, 9, 36
    w = synth.Workload(86, R, J, B, v=V, n_blocks=J * B - 3)
    kinds = [KINDS[(r * 5) % len(KINDS)] for r in range(R)]
    for r, k in enumerate(kinds):
        _tamper(w, r, k)
    want = _oracle_all(w, J, B)
    packed = [BT.pack_headers(w.headers[r]) for r in range(R)]
    bt = BT.Batcher(J, B, V, max_requests=8, n_lanes=2)
    for rep in range(3):
        bt.cork()
        form = [(r + rep) % 3 != 0 for r in range(R)]                       # two thirds packed, rotating
        tickets = [bt.submit_header_range(w.input48(r), packed[r] if form[r] else w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r],
                                          w.trusted[r], packed=form[r]) for r in range(R)]
        bt.cork(False)
        for r in range(R):
            rc, (out, res) = bt.wait(tickets[r], allow=tuple(range(1, 10)))
            assert rc == want[r][0], (rep, r, kinds[r], form[r], rc, want[r][0], _lib.last_error())
            if rc != T.ERR_BAD_HEADER:
                assert out == want[r][1] and res.tobytes() == want[r][2].tobytes(), (rep, r, kinds[r], form[r])
    # an inconsistent block is refused at submit (no ticket): offsets running backwards / beyond the block / truncated
    bad = packed[0].copy()
    bad[8 + 4 * 3:8 + 4 * 3 + 4] = np.frombuffer((5).to_bytes(4, "little"), np.uint8)
    for blk in (bad, packed[0][:200], np.zeros(64, np.uint8)):
        with pytest.raises(_lib.BsxError) as e:
            bt.submit_header_range(w.input48(0), blk, int(w.first_height[0]), int(w.latest[0]), w.validators[0], w.trusted[0], packed=True)
        assert e.value.status == T.ERR_BAD_HEADER
    bt.close()


def test_synchronous_packed_call_with_and_without_coalescing():
    """bsx_header_range_packed: submit + wait on a coalescing context, host-side unpack + the ordinary path on a plain one."""
    J, B, V = 4, 16, 6
    w = synth.Workload(87, 3, J, B, v=V)
    _tamper(w, 1, "sig")
    want = _oracle_all(w, J, B)
    L = _lib.lib()
    cid = np.frombuffer(b"celestia", np.uint8).copy()

    def call(r):
        pk = BT.pack_headers(w.headers[r])
        out, res = np.zeros(64, np.uint8), np.zeros(1, T.COMMIT_RESULT)
        inp = np.frombuffer(w.input48(r), np.uint8).copy()
        rc = L.bsx_header_range_packed(_lib.context(0), C.c_uint32(J), C.c_uint32(B), _lib.p(inp), _lib.p(pk), C.c_uint64(pk.size),
                                       C.c_uint64(int(w.first_height[r])), C.c_uint64(int(w.latest[r])), _lib.p(np.ascontiguousarray(w.validators[r])),
                                       _lib.p(np.ascontiguousarray(w.trusted[r])), C.c_uint32(V), _lib.p(cid), C.c_uint32(8), _lib.p(out), _lib.p(res))
        return rc, out.tobytes(), res[0]
    for coalesce in (False, True):
        if coalesce:
            BT.enable_coalescing(J, B, V)
        try:
            for r in range(3):
                rc, out, res = call(r)
                assert rc == want[r][0] and out == want[r][1] and res.tobytes() == want[r][2].tobytes(), (coalesce, r)
        finally:
            if coalesce:
                BT.disable_coalescing()


def test_page_locked_callers_upload_from_where_the_headers_lie():
    """BSX_SUBMIT_INPUTS_STAY with headers in page-locked memory — hipHostMalloc'ed (torch pin_memory) and bsx_host_register'ed pageable
    memory — and in plain pageable memory (staged as before): same results; the registered buffer is REUSED for different ranges."""
    import torch
    J, B, V, R = 4, 16, 6, 6
    w = synth.Workload(88, R, J, B, v=V)
    want = _oracle_all(w, J, B)
    L = _lib.lib()
    bt = BT.Batcher(J, B, V, max_requests=4)
    nb = w.headers[0].nbytes
    pinned = torch.empty(nb, dtype=torch.uint8, pin_memory=True).numpy().view(T.HEADER)
    reg = np.zeros(w.headers[0].size, T.HEADER)
    _lib.check(L.bsx_host_register(_lib.context(0), _lib.p(reg), C.c_uint64(reg.nbytes)))
    try:
        for r in range(R):
            buf = (pinned, reg, np.ascontiguousarray(w.headers[r]))[r % 3]
            buf[:] = w.headers[r]
            t = bt.submit_header_range(w.input48(r), buf, int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r], inputs_stay=True)
            rc, (out, res) = bt.wait(t)
            assert rc == want[r][0] == T.OK and out == want[r][1] and res.tobytes() == want[r][2].tobytes(), r
    finally:
        bt.close()
        _lib.check(L.bsx_host_unregister(_lib.context(0), _lib.p(reg)))


def test_tickets_nobody_waits_for_polling_and_destroy_with_requests_pending():
    """Round 6 completion: a waiter copies its own results out; tickets nobody waits for are taken out by the worker (results land at the
    caller's pointers all the same); bsx_poll reports done only with the outputs in place; bsx_batcher_destroy with requests still
    collecting fails them with a status instead of leaving their waiters hanging."""
    import time
    J, B, V, R = 4, 16, 6, 12
    w = synth.Workload(89, R, J, B, v=V)
    _tamper(w, 5, "sig")
    want = _oracle_all(w, J, B)
    bt = BT.Batcher(J, B, V, max_requests=4)
    tickets = [bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]) for r in range(R)]
    # fire and forget half of them: drop the Ticket objects — the Batcher keeps their output arrays alive and the worker fills them
    outs = {r: tickets[r].outputs for r in range(0, R, 2)}
    numbers = {r: tickets[r].ticket for r in range(0, R, 2)}
    for r in range(0, R, 2):
        tickets[r] = None
    t0 = time.time()
    probe = BT.Ticket("header_range", numbers[R - 2], outs[R - 2])
    while not bt.done(probe):
        assert time.time() - t0 < 30
        time.sleep(0.001)
    for r in range(0, R, 2):                     # done, never waited for: outputs are in place
        p = BT.Ticket("header_range", numbers[r], outs[r])
        t1 = time.time()
        while not bt.done(p):
            assert time.time() - t1 < 30
        assert outs[r][0].tobytes() == want[r][1] and outs[r][1][0].tobytes() == want[r][2].tobytes(), r
    for r in range(1, R, 2):                     # the others: waited for twice (a ticket may be)
        for _ in range(2):
            rc, (out, res) = bt.wait(tickets[r], allow=tuple(range(1, 10)))
            assert rc == want[r][0] and out == want[r][1] and res.tobytes() == want[r][2].tobytes(), r
    # destroy with a corked, half-full batch and two threads parked in bsx_wait: what was collected RUNS (the worker sees the stop), the
    # waiters are released with their results, and nobody is left on a deleted batcher
    bt.cork()
    pend = [bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r]) for r in range(2)]
    got = []
    th = [threading.Thread(target=lambda tk=tk: got.append(_lib.lib().bsx_wait(bt.h, C.c_uint64(tk.ticket)))) for tk in pend]
    for x in th:
        x.start()
    time.sleep(0.05)
    t0 = time.time()
    bt.close()
    for x in th:
        x.join(timeout=30)
        assert not x.is_alive(), "a waiter was left hanging by bsx_batcher_destroy"
    assert time.time() - t0 < 10 and got == [T.OK, T.OK], got
    for r, tk in enumerate(pend):
        assert tk.outputs[0].tobytes() == want[r][1] and tk.outputs[1][0].tobytes() == want[r][2].tobytes(), r


def test_sixteen_native_callers_on_two_cpus():
    """the batcher's workers spin while a batch collects (tens of microseconds) — what if the box grants fewer CPUs
    than there are callers + workers?  16 native caller threads (tests/hostcheck/concurrent_driver.cpp) + the lanes' workers pinned to TWO
    CPUs: every call still returns the chain's own target hash, the run ends on time, and calls keep completing (no starvation of the
    threads the spinning workers wait for)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpus = sorted(os.sched_getaffinity(0))[:2]
    code = (
        "import os, sys, json\n"
        f"os.sched_setaffinity(0, {set(cpus)!r})\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch\n"
        "from bench_legs.latency import concurrent_leg\n"
        "r = concurrent_leg(torch.device('cuda:0'), 8, 32, 20, ks=(16,), seconds=0.5, serial=False, forms=('packed',), form_ks=(16,))\n"
        "print('ROWS ' + json.dumps({'rec': r['coalesced_shared_context'][0], 'packed': r['coalesced_packed_headers'][0]}))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]          # concurrent_leg asserts every call's status and output itself
    import json
    rows = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("ROWS ")][-1][5:])
    for name, row in rows.items():
        assert row["threads"] == 16 and row["calls"] >= 16 * 20, (name, row)     # >= 20 calls per thread in 0.5 s on two CPUs
        assert row["p99_ms"] < 250, (name, row)
