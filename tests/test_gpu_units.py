"""GPU suite (round 4): the Goldilocks witness of the skip / step verification — COMMIT, SKIP and STEP units
(include/bsx_layout.h; builder.skip / builder.step, circuits/header_range.rs:42-48, circuits/next_header.rs:32-36) — through the C
ABI, element for element against the oracle: host tier (bsx_header_range, bsx_next_header, bsx_verify_commits), the batched
pipeline (BSX_PIPE_WITNESS / BSX_PIPE_CAPS over the units, un-joined steps, streamed inputs, sharded ranks) and mode S
(bsx_dev_verify_commits' compact COMMIT units + expansion)."""
import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import types as T

pytestmark = pytest.mark.gpu

VALID_WITNESS_RC = (T.OK, T.ERR_ASSERT, T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER)


def first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return (int(d[0]), int(a[d[0]]), int(b[d[0]]), d.size) if d.size else None


def unit_slices(J, B, V):
    ml, rl, cl, sl = T.map_layout(B), T.reduce_layout(), T.commit_layout(V), T.skip_layout(V)
    a = J * int(ml["n_elements"]) + (J - 1) * int(rl["n_elements"])
    return a, a + int(cl["n_elements"]), a + int(cl["n_elements"]) + int(sl["n_elements"])


@pytest.mark.parametrize("J,B,v,v_max", [(2, 8, 10, 10), (4, 16, 100, 100), (2, 4, 512, 512), (2, 4, 5, 8), (1, 2, 1, 1), (2, 2, 3, 3)])
def test_header_range_units_vs_oracle(J, B, v, v_max):
    from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
    w = synth.Workload(900 + v, 2, J, B, v=v, v_max=v_max, absent_permille=120 if v >= 10 else 0, nil_permille=60 if v >= 10 else 0)
    circ = CombinedSkipCircuit(v_max, J, B)
    a, b, c = unit_slices(J, B, v_max)
    for r in range(2):
        S = int(w.first_height[r])
        f = InputDataFetcher(w.headers[r], S, int(w.latest[r]))
        rc, ref_out, ref_res, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], S, int(w.latest[r]), w.validators[r], w.trusted[r],
                                                       want_witness=True)
        out, res, wit = circ.prove(w.input48(r), f, w.validators[r], w.trusted[r], want_witness=True, allow=VALID_WITNESS_RC)
        assert circ.last_rc == rc and out == ref_out
        want = oracle.expand_range_witness(J, B, cw, v_max=v_max)
        assert wit.size == want.size == c
        assert first_diff(wit[a:b], want[a:b]) is None, ("COMMIT unit", first_diff(wit[a:b], want[a:b]))
        assert first_diff(wit[b:c], want[b:c]) is None, ("SKIP unit", first_diff(wit[b:c], want[b:c]))
        assert (wit == want).all()


def test_failing_skips_carry_the_oracles_witness():
    """tampered requests (signature, message, chain id, trusted hash, voting powers, trusted set): same status, same witness — the
    assertion bools say why"""
    from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
    J, B, V = 2, 4, 12
    w = synth.Workload(931, 1, J, B, v=V)
    S = int(w.first_height[0])
    f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
    a, b, c = unit_slices(J, B, V)
    seen = set()

    def both(inp, tv, rv, cid=b"celestia"):
        circ = CombinedSkipCircuit(V, J, B, chain_id=cid)
        rc, ref_out, _, cw = oracle.header_range(J, B, inp, w.headers[0], S, int(w.latest[0]), tv, rv, want_witness=True, chain_id=cid)
        assert rc in VALID_WITNESS_RC
        out, _, wit = circ.prove(inp, f, tv, rv, want_witness=True, allow=VALID_WITNESS_RC)
        assert circ.last_rc == rc and out == ref_out
        want = oracle.expand_range_witness(J, B, cw, v_max=V)
        assert first_diff(wit[a:], want[a:]) is None, (rc, first_diff(wit[a:], want[a:]))
        assert (wit == want).all()
        seen.add(rc)
    tv, rv, inp = w.validators[0], w.trusted[0], w.input48(0)
    both(inp, tv, rv)
    t = tv.copy(); t[3]["signature"][10] ^= 2; both(inp, t, rv)
    t = tv.copy(); t[5]["message"][20] ^= 1; both(inp, t, rv)
    both(inp, tv, rv, cid=b"mocha-4")
    bad = bytearray(inp); bad[9] ^= 1; both(bytes(bad), tv, rv)
    t = tv.copy(); t["is_signed"][:9] = 0; both(inp, t, rv)                      # 2/3 missed
    r = rv.copy(); r[0]["voting_power"] += 1; both(inp, tv, r)                   # trusted set no longer hashes to the header's field
    r = rv.copy(); r["enabled"][6:] = 0; both(inp, tv, r)
    t = tv.copy(); t[1]["message_len"] = 40; both(inp, t, rv)
    assert {T.OK, T.ERR_ASSERT, T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER} <= seen


@pytest.mark.parametrize("v,v_max", [(100, 100), (7, 8), (33, 64), (512, 512), (1, 1)])
def test_next_header_units_vs_oracle(v, v_max):
    from blobstreamx_amd.builder import CombinedStepCircuit
    w = synth.Workload(960 + v, 1, 1, 8, v=v, v_max=v_max, mode="S", absent_permille=100 if v > 8 else 0)
    circ = CombinedStepCircuit(v_max)
    S = int(w.first_height[0])
    ncl = int(T.commit_layout(v_max)["n_elements"])
    for k in range(3):
        inp = (S + k).to_bytes(8, "big") + w.hashes[0, k].tobytes()
        vals = w.validators[k]
        for latest in (int(w.latest[0]), S + k + 2):                            # the second clamps data_hash_proofs[0] away (A10 fails)
            rc, want_out, wcr, cw = oracle.next_header(inp, w.headers[0, k], w.headers[0, k + 1], latest, vals, want_witness=True)
            assert rc in VALID_WITNESS_RC
            out, cr, wit = circ.prove(inp, w.headers[0, k], w.headers[0, k + 1], latest, vals, want_witness=True, allow=VALID_WITNESS_RC)
            assert circ.last_rc == rc and out == want_out and cr.tobytes() == wcr.tobytes()
            want = oracle.expand_step_witness(v_max, cw)
            assert wit.size == want.size == T.next_header_witness_elements(v_max)
            assert first_diff(wit[:ncl], want[:ncl]) is None, ("COMMIT unit", first_diff(wit[:ncl], want[:ncl]))
            assert first_diff(wit[ncl:], want[ncl:]) is None, ("STEP unit", first_diff(wit[ncl:], want[ncl:]))
    # tampering of every link the step enforces: same status, same witness
    k = 1
    inp = (S + k).to_bytes(8, "big") + w.hashes[0, k].tobytes()
    vals = w.validators[k]

    def both(i, ph, nh, vv, cid=b"celestia"):
        c2 = CombinedStepCircuit(v_max, chain_id=cid)
        rc, want_out, _, cw = oracle.next_header(i, ph, nh, int(w.latest[0]), vv, chain_id=cid, want_witness=True)
        if rc not in VALID_WITNESS_RC:
            with pytest.raises(_lib.BsxError) as ei:
                c2.prove(i, ph, nh, int(w.latest[0]), vv, want_witness=True)
            assert ei.value.status == rc
            return rc
        out, _, wit = c2.prove(i, ph, nh, int(w.latest[0]), vv, want_witness=True, allow=VALID_WITNESS_RC)
        assert c2.last_rc == rc and out == want_out
        want = oracle.expand_step_witness(v_max, cw)
        assert first_diff(wit, want) is None, (rc, first_diff(wit, want))
        return rc
    ph, nh = w.headers[0, k], w.headers[0, k + 1]
    both(inp, ph, nh, vals, cid=b"mocha-4")
    h2 = nh.copy(); h2["last_block_id"][5] ^= 1; both(inp, ph, h2, vals)
    h2 = ph.copy(); h2["hash"][3][7] ^= 1; both(inp, h2, nh, vals)               # prev.next_validators_hash
    h2 = nh.copy(); h2["hash"][2][9] ^= 1; both(inp, ph, h2, vals)               # next.validators_hash
    bad = bytearray(inp); bad[3] ^= 1; both(bytes(bad), ph, nh, vals)            # height
    bad = bytearray(inp); bad[20] ^= 1; both(bytes(bad), ph, nh, vals)           # prev hash
    if v > 1:
        t = vals.copy(); t[0]["signature"][1] ^= 1; both(inp, ph, nh, t)


@pytest.mark.parametrize("n,v,v_max", [(5, 33, 64), (3, 100, 100), (70, 12, 16)])
def test_verify_commits_witness_vs_oracle(n, v, v_max):
    from blobstreamx_amd.builder import verify_commits
    w = synth.Workload(990 + n, 1, 1, max(8, 1 << (n - 1).bit_length()), v=v, v_max=v_max, mode="S", absent_permille=150, nil_permille=80)
    vals = w.validators[:n].copy()
    vals[1][2]["signature"][5] ^= 1
    hh = w.hashes[0, 1:n + 1]
    res, ok, wit = verify_commits(vals, hh, want_witness=True)
    lay = T.commit_layout(v_max)
    for c in range(n):
        wres, wok, cw = oracle.verify_commit(vals[c], hh[c].tobytes(), want_witness=True)
        assert res[c].tobytes() == wres.tobytes() and (ok[c] == wok).all()
        want = oracle.expand_witness(lay, 1, cw)
        assert first_diff(wit[c], want) is None, (c, first_diff(wit[c], want))


def _oracle_units(w, r, J, B, V):
    rc, out, cres, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r],
                                            w.trusted[r], want_witness=True)
    full = oracle.expand_range_witness(J, B, cw, v_max=V)
    a, b, c = unit_slices(J, B, V)
    return rc, out, full[a:b], full[b:c]


@pytest.mark.parametrize("J,B,V,R,E,streaming", [(2, 8, 10, 8, 1, False), (4, 8, 24, 16, 2, False), (2, 8, 100, 16, 2, True), (2, 4, 5, 4, 1, True)])
def test_pipeline_unit_witness_vs_oracle(J, B, V, R, E, streaming):
    """BSX_PIPE_WITNESS | BSX_PIPE_COMMIT: the COMMIT / SKIP units of every owned range after several un-joined steps, keyed and
    generic Ed25519, resident and streamed inputs; tampered ranges included (a failing skip still has its witness)."""
    from blobstreamx_amd.engine import Pipeline
    w = synth.Workload(1200 + V, R, J, B, v=V, absent_permille=100 if V >= 10 else 0)
    w.validators[1][0]["signature"][3] ^= 1
    w.trusted[2][1]["voting_power"] += 5
    p = Pipeline(J, B, V, R, n_chunks=E, with_witness=True, with_commit=True)
    p.upload_workload(w)
    if streaming:
        p.enable_input_streaming(True)
    for _ in range(3):
        p.step()
    res = p.download()
    Rc = R // E
    nm = J * int(T.map_layout(B)["n_elements"])
    for e in range(E):
        cu, su = p.unit_witness_numpy(e)
        wm, wr, _ = p.witness_numpy(e)
        for k in range(Rc):
            r = e * Rc + k
            rc, out, wc, ws = _oracle_units(w, r, J, B, V)
            if k in (0, Rc - 1):          # the streamed-input path had no correctness test — map-job witness too
                _, _, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r],
                                                  w.trusted[r], want_witness=True)
                full = oracle.expand_range_witness(J, B, cw)
                assert first_diff(wm[k * nm:(k + 1) * nm], full[:nm]) is None, (r, "map jobs")
                assert first_diff(wr[k * (full.size - nm):(k + 1) * (full.size - nm)], full[nm:]) is None, (r, "reduce nodes")
            assert res["output64"][r].tobytes() == out
            want_skip = rc if rc in (T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER, T.ERR_ASSERT) else T.OK
            assert int(res["skip_status"][r]) == want_skip or int(res["range_status"][r]) != 0, (r, rc, res["skip_status"][r])
            assert first_diff(cu[k], wc) is None, (r, "COMMIT", first_diff(cu[k], wc))
            assert first_diff(su[k], ws) is None, (r, "SKIP", first_diff(su[k], ws))
    p.close()


def test_pipeline_unit_caps_vs_oracle():
    """BSX_PIPE_CAPS over the COMMIT / SKIP units (no 64x image): Poseidon trees of the units hashed straight from the compact bytes
    = the oracle's Poseidon over the oracle's expanded units"""
    from blobstreamx_amd.engine import Pipeline
    J, B, V, R = 2, 8, 10, 8
    w = synth.Workload(1300, R, J, B, v=V)
    p = Pipeline(J, B, V, R, n_chunks=1, with_witness=False, with_commit=True, with_caps=True, leaf_len=135, cap_height=2)
    p.upload_workload(w)
    p.step(); p.step()
    tc, ts = p.unit_caps_numpy(0)
    for r in (0, 3, 7):
        rc, out, wc, ws = _oracle_units(w, r, J, B, V)
        assert rc == T.OK
        for tree, unit in ((tc[r], wc), (ts[r], ws)):
            n_leaves = 1
            while n_leaves * 135 < unit.size:
                n_leaves *= 2
            ch = min(2, n_leaves.bit_length() - 1)
            want, _ = oracle.poseidon_merkle_tree(unit, 135, n_leaves, ch)
            assert tree.shape == want.shape and (tree == want).all()
    p.close()


def test_sharded_ranks_emit_the_owners_units():
    """world 2 / 4 emulated on one GPU: every rank's pipeline (own job slice, all-gather delivered by a callback) leaves the COMMIT
    / SKIP units of the ranges it OWNS, equal to the oracle's"""
    from blobstreamx_amd.engine import PipelinedEngines, run_world_on_one_gpu
    for world, J, B, V, R, E in ((2, 4, 8, 10, 4, 1), (4, 8, 8, 24, 2, 2)):
        w = synth.Workload(1400 + world, world * R, J, B, v=V)
        engs = [PipelinedEngines(J, B, V, R, n_engines=E, rank=g, world=world) for g in range(world)]
        for e in engs:
            e.upload_workload(w)
        run_world_on_one_gpu(engs)
        for g, e in enumerate(engs):
            res = e.download()
            for c in range(E):
                cu, su = e.unit_witness_numpy(c)
                for k in range(e.Rc):
                    r = g * R + c * e.Rc + k
                    rc, out, wc, ws = _oracle_units(w, r, J, B, V)
                    kk = c * e.Rc + k
                    assert rc == T.OK and res["output64"][kk].tobytes() == out and int(res["skip_status"][kk]) == 0 and int(res["range_status"][kk]) == 0
                    assert first_diff(cu[k], wc) is None and first_diff(su[k], ws) is None, (world, g, r)
        for e in engs:
            e.close()


@pytest.mark.parametrize("n_commits,v,v_max,world", [(64, 100, 100, 1), (32, 40, 64, 2), (16, 512, 512, 1)])
def test_mode_s_commit_units_vs_oracle(n_commits, v, v_max, world):
    """mode S (BASELINE config #5's 'bit-exact witness diff'): bsx_dev_verify_commits leaves every commit's COMMIT unit, expanded on
    the same stream; sampled commits of every rank's slice against the oracle, twice (the resident buffers are rewritten in place)"""
    import torch
    from blobstreamx_amd.stress import CommitShard
    w = synth.Workload(1500 + v, 1, 1, n_commits, v=v, v_max=v_max, mode="S", absent_permille=100, nil_permille=50)
    vals = w.validators[:n_commits].copy()
    vals[3][1]["signature"][9] ^= 1
    hh = w.hashes[0, 1:n_commits + 1]
    lay = T.commit_layout(v_max)
    for g in range(world):
        sh = CommitShard(n_commits, v_max, rank=g, world=world, expand=True)
        sh.upload(vals, hh)
        for it in range(2):
            sh.step()
            pick = sorted({0, 3 % sh.n, sh.n // 2, sh.n - 1})
            got = sh.witness_of(pick)
            for i, c in enumerate(pick):
                gc = sh.first + c
                wres, wok, cw = oracle.verify_commit(vals[gc], hh[gc].tobytes(), want_witness=True)
                want = oracle.expand_witness(lay, 1, cw)
                assert first_diff(got[i], want) is None, (g, it, gc, first_diff(got[i], want))
        ok, res, fold = sh.download()
        assert fold["n_commits"] == sh.n
        del sh
        torch.cuda.empty_cache()


def test_capacity_checked_entry_points_refuse_a_round_3_sized_buffer():
    """the witnesses of bsx_header_range / bsx_next_header / bsx_verify_commits grew in round 4 (COMMIT + SKIP / STEP units
    appended).  The *_cap forms take the buffer's capacity and fail with BSX_ERR_BAD_ARG before anything runs when it is the round-3
    size (map jobs + reduce nodes only) — instead of overrunning the host buffer."""
    import ctypes as C
    from blobstreamx_amd import _lib
    J, B, V = 2, 8, 5
    w = synth.Workload(77, 1, J, B, v=V)
    L, ctx = _lib.lib(), _lib.context(0)
    ml, rl = T.map_layout(B), T.reduce_layout()
    old = J * int(ml["n_elements"]) + (J - 1) * int(rl["n_elements"])          # what a round-3 host allocated
    need = T.header_range_witness_elements(J, B, V)
    assert old < need
    wit = np.zeros(need, np.uint64)
    out, res = np.zeros(64, np.uint8), np.zeros(1, T.COMMIT_RESULT)
    inp = np.frombuffer(w.input48(0), np.uint8).copy()
    hdr, tv, rv = np.ascontiguousarray(w.headers[0]), np.ascontiguousarray(w.validators[0]), np.ascontiguousarray(w.trusted[0])
    cid = np.frombuffer(b"celestia", np.uint8).copy()

    def call(cap):
        return L.bsx_header_range_cap(ctx, C.c_uint32(J), C.c_uint32(B), _lib.p(inp), _lib.p(hdr), C.c_uint64(int(w.first_height[0])), C.c_uint64(hdr.size),
                                      C.c_uint64(int(w.latest[0])), _lib.p(tv), _lib.p(rv), C.c_uint32(V), _lib.p(cid), C.c_uint32(8), _lib.p(out), _lib.p(res),
                                      _lib.p(wit), C.c_uint64(cap))
    assert call(old) == T.ERR_BAD_ARG and "needs" in _lib.last_error() and not wit.any()
    assert call(need) == T.OK and wit.any()
    hh = np.ascontiguousarray(w.hashes[0, w.n_blocks]).reshape(1, 32)
    cw = np.zeros(int(T.commit_layout(V)["n_elements"]), np.uint64)
    ok = np.zeros(V, np.uint8)
    assert L.bsx_verify_commits_cap(ctx, _lib.p(tv), C.c_uint32(1), C.c_uint32(V), _lib.p(hh), _lib.p(res), _lib.p(ok), _lib.p(cw), C.c_uint64(cw.size - 1)) == T.ERR_BAD_ARG
    assert L.bsx_verify_commits_cap(ctx, _lib.p(tv), C.c_uint32(1), C.c_uint32(V), _lib.p(hh), _lib.p(res), _lib.p(ok), _lib.p(cw), C.c_uint64(cw.size)) == T.OK
