"""GPU suite (`-m gpu`): the HIP path, called through the C ABI (libbsx.so), against
  (1) the golden vectors derived from the reference's fixtures (tests/golden/mocha4.json), and
  (2) the CPU oracle (oracle/) on seeded synthetic inputs — bit-exact for every byte, record, status and
      Goldilocks witness element.
Mirrors the reference's tests test_get_data_commitment / test_prove_header_chain / test_encode_data_root_tuple
(circuits/builder.rs:488-608) and the header_range / next_header end-to-end shapes (circuits/header_range.rs:193-266,
circuits/next_header.rs:130-179)."""
import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import types as T
from blobstreamx_amd.builder import CombinedSkipCircuit, CombinedStepCircuit, DataCommitmentBuilder, InputDataFetcher, verify_commits

pytestmark = pytest.mark.gpu


def rec_bytes(r):
    r = np.array(r, dtype=T.SUBCHAIN).copy()
    r["_pad"] = 0
    return r.tobytes()


def res_bytes(r):
    r = np.array(r, dtype=T.COMMIT_RESULT).copy()
    r["_pad"] = 0
    return r.tobytes()


@pytest.fixture(scope="module")
def builder():
    return DataCommitmentBuilder()


@pytest.fixture(scope="module")
def fetcher(mocha):
    return InputDataFetcher(mocha["headers"], mocha["first_height"], mocha["latest"])


# ------------------------------------------------------------------ golden vectors (reference fixtures)
def test_golden_header_hashes_and_proofs(golden, mocha, fetcher):
    hashes, dh, lb = fetcher.get_inclusion_proofs()
    for i, h in enumerate(mocha["heights"]):
        b = golden["blocks"][str(h)]
        assert hashes[i].tobytes().hex() == b["header_hash"]
        assert [bytes(a).hex() for a in dh[i]["aunts"]] == b["data_hash_proof"]["aunts"]
        assert bytes(dh[i]["leaf"]).hex() == b["data_hash_proof"]["leaf"]
        assert [bytes(a).hex() for a in lb[i]["aunts"]] == b["last_block_id_proof"]["aunts"]
        assert bytes(lb[i]["leaf"]).hex() == b["last_block_id_proof"]["leaf"]


def test_golden_encode_data_root_tuple(golden, builder):
    k = golden["kats"]["encode_data_root_tuple"]          # circuits/builder.rs:584-605
    assert builder.encode_data_root_tuple(bytes.fromhex(k["data_hash"]), k["height"]).hex() == k["expected"]


def test_golden_get_data_commitment(golden, mocha, fetcher, builder):
    for name, want in golden["data_commitments"].items():  # circuits/builder.rs:488-529, MAX_LEAVES = 4
        s, e = map(int, name.split("-"))
        inp = fetcher.get_data_commitment_inputs(s, e, 4)
        assert inp["expected_data_commitment"].hex() == want
        dhs = np.stack([p["leaf"][2:34] for p in inp["data_hash_proofs"]])
        assert builder.get_data_commitment(dhs, s, e).hex() == want


def test_golden_prove_header_chain(golden, mocha, fetcher, builder):
    inp = fetcher.get_data_commitment_inputs(10000, 10004, 4)   # circuits/builder.rs:531-564
    assert inp["start_header_hash"] == mocha["hashes"][0] and inp["end_header_hash"] == mocha["hashes"][4]
    rec, _ = builder.prove_subchain(inp, 10000, 10004, 10004, inp["end_header_hash"])
    assert rec["assert_fail"] == 0 and rec["is_enabled"] == 1
    assert bytes(rec["data_merkle_root"]).hex() == golden["data_commitments"]["10000-10004"]
    assert bytes(rec["end_header"]) == mocha["hashes"][4] and rec["end_block"] == 10004


def test_golden_prove_data_commitment_shapes(golden, mocha, fetcher, builder):
    hh = mocha["hashes"]
    for (J, B, s, e) in [(2, 2, 10000, 10004), (4, 4, 10000, 10002), (2, 8, 10002, 10004), (1, 4, 10000, 10004),
                         (4, 1, 10000, 10004), (8, 2, 10000, 10001)]:
        out = builder.prove_data_commitment(fetcher, J, B, s, hh[s - 10000], e, hh[e - 10000], want_witness=True)
        assert out["data_commitment"].hex() == golden["data_commitments"][f"{s}-{e}"], (J, B, s, e)
        ctx = oracle.make_ctx(s, hh[s - 10000], e, hh[e - 10000])
        rc, ref = oracle.prove_data_commitment(J, B, ctx, mocha["headers"], 10000, mocha["latest"], want_witness=True)
        assert rc == T.OK
        assert [rec_bytes(r) for r in out["records"]] == [rec_bytes(r) for r in ref["records"]]
        assert (out["witness"] == oracle.expand_range_witness(J, B, ref["compact"])).all(), (J, B, s, e)


def test_golden_next_header(golden, mocha, fetcher, builder):
    dc = builder.prove_next_header_data_commitment(fetcher, 10000, mocha["hashes"][0], 10001)   # builder.rs:411-443
    assert dc.hex() == golden["data_commitments"]["10000-10001"]
    with pytest.raises(_lib.BsxError) as ei:
        builder.prove_next_header_data_commitment(fetcher, 10000, mocha["hashes"][1], 10001)
    assert ei.value.status == T.ERR_ASSERT


def test_golden_commits(golden, mocha):
    vals = np.stack(mocha["commits"])
    res, ok = verify_commits(vals, np.frombuffer(b"".join(mocha["hashes"]), np.uint8).reshape(5, 32))
    for i, h in enumerate(mocha["heights"]):
        b = golden["blocks"][str(h)]
        assert bytes(res[i]["validators_hash"]).hex() == b["validators_hash"]
        assert res[i]["n_signed"] == 2 and res[i]["n_bad_signature"] == 0 and res[i]["n_bad_message"] == 0
        assert res[i]["two_thirds_ok"] == 1 and res[i]["signed_power"] == 50_000_000
        assert list(ok[i]) == [1, 1, 0, 0]
        ref, rok = oracle.verify_commit(mocha["commits"][i], mocha["hashes"][i])
        assert res_bytes(res[i]) == res_bytes(ref) and (ok[i] == rok).all()


def test_golden_reject_paths(mocha, fetcher, builder):
    hh = mocha["hashes"]
    for (J, B, s, sh, e, eh) in [(2, 2, 10000, hh[0], 10004, hh[3]), (2, 2, 10000, hh[1], 10004, hh[4]),
                                 (1, 2, 10000, hh[0], 10004, hh[4])]:
        out = builder.prove_data_commitment(fetcher, J, B, s, sh, e, eh, raise_on_assert=False)
        rc, ref = oracle.prove_data_commitment(J, B, oracle.make_ctx(s, sh, e, eh), mocha["headers"], 10000, mocha["latest"])
        assert out["rc"] == rc == T.ERR_ASSERT
        assert out["result"]["assert_fail"] == ref["status"]
        assert [rec_bytes(r) for r in out["records"]] == [rec_bytes(r) for r in ref["records"]]


# ------------------------------------------------------------------ synthetic vs oracle
def random_headers(n, seed):
    """headers with edge-case field lengths: empty fields, long chain ids, first-block last_block_id, 73-byte block id"""
    rnd = np.random.default_rng(seed)
    out = np.zeros(n, T.HEADER)
    for i in range(n):
        f = []
        f.append(bytes([0x08, 11, 0x10, 1]) if i % 7 else b"")
        cid = bytes(rnd.integers(97, 123, size=int(rnd.integers(0, 51)), dtype=np.uint8))
        f.append((b"\x0a" + bytes([len(cid)]) + cid) if cid else b"")
        f.append(b"\x08" + bytes(rnd.integers(1, 128, size=int(rnd.integers(1, 10)), dtype=np.uint8) | 0x80)[:-1] + b"\x01")
        f.append(bytes(rnd.integers(0, 256, size=int(rnd.integers(0, 21)), dtype=np.uint8)))
        kind = i % 5
        lbi = {0: 72, 1: 72, 2: 2, 3: 73, 4: 76}[kind] if i % 11 else 0
        f.append(bytes(rnd.integers(0, 256, size=lbi, dtype=np.uint8)))
        for j in range(8):
            ln = 34 if (i + j) % 13 else 0
            f.append(bytes(rnd.integers(0, 256, size=ln, dtype=np.uint8)))
        f.append(bytes(rnd.integers(0, 256, size=int(rnd.integers(0, 25)), dtype=np.uint8)))
        out[i] = T.pack_header(f)
    return out


def test_header_hashes_vs_oracle_edge_lengths():
    hdrs = random_headers(700, 3)
    f = InputDataFetcher(hdrs, 1, 10_000)
    got = f.header_hashes()
    want = oracle.header_hash_only(hdrs)
    assert (got == want).all()


def test_header_hashes_garbage_beyond_len_is_ignored():
    w = synth.Workload(5, 1, 2, 8, v=3)
    hdrs = w.headers[0].copy()
    want = oracle.header_hash_only(hdrs)
    raw = hdrs.view(np.uint8).reshape(-1, 512)
    dirty = raw.copy()
    # fill the unused tail of every field with 0xEE
    offs = [16, 40, 92, 104, 124] + [200 + 36 * j for j in range(8)] + [488]
    caps = T.HEADER_FIELD_CAP
    for i in range(dirty.shape[0]):
        for k in range(14):
            ln = int(raw[i, k])
            dirty[i, offs[k] + ln:offs[k] + caps[k]] = 0xEE
    got = InputDataFetcher(dirty.view(T.HEADER).reshape(-1), 1, 100).header_hashes()
    assert (got == want).all()


def test_bad_header_is_an_error_not_a_crash():
    hdrs = random_headers(3, 9)
    hdrs[1]["len"][5] = 40   # > capacity 36
    with pytest.raises(_lib.BsxError) as ei:
        InputDataFetcher(hdrs, 1, 100).header_hashes()
    assert ei.value.status == T.ERR_BAD_HEADER


@pytest.mark.parametrize("B,latest_off,span", [(4, 2, 4), (4, 0, 4), (8, -3, 8), (8, -20, 8), (16, 2, 5), (2, 2, 0), (1, 2, 1)])
def test_data_commitment_inputs_vs_oracle(B, latest_off, span):
    w = synth.Workload(11, 1, 2, 16, v=2)
    S = int(w.first_height[0])
    start = S + 3
    latest = start + B + 2 + latest_off
    f = InputDataFetcher(w.headers[0], S, latest)
    got = f.get_data_commitment_inputs(start, start + span, B)
    rc, ref = oracle.data_commitment_inputs(w.headers[0], S, latest, start, start + span, B)
    assert rc == T.OK
    assert got["start_header_hash"] == ref["start_header"] and got["end_header_hash"] == ref["end_header"]
    assert got["data_hash_proofs"].tobytes() == ref["data_hash_proofs"].tobytes()
    assert got["last_block_id_proofs"].tobytes() == ref["last_block_id_proofs"].tobytes()
    assert got["expected_data_commitment"] == ref["expected_data_commitment"]


def test_data_commitment_inputs_range_too_long():
    w = synth.Workload(11, 1, 2, 16, v=2)
    S = int(w.first_height[0])
    with pytest.raises(_lib.BsxError) as ei:
        InputDataFetcher(w.headers[0], S, S + 40).get_data_commitment_inputs(S, S + 5, 4)   # circuits/input.rs:154
    assert ei.value.status == T.ERR_RANGE_TOO_LONG


@pytest.mark.parametrize("B", [1, 2, 4, 32, 64, 256])
def test_prove_subchain_vs_oracle(B, builder):
    w = synth.Workload(20 + B, 1, 1, B, v=2)
    S = int(w.first_height[0])
    hdrs, hs = w.headers[0], w.hashes[0]
    f = InputDataFetcher(hdrs, S, S + B + 2)
    inp = f.get_data_commitment_inputs(S, S + B, B)
    # global end positions: at the batch end, in the middle, at the first block, before the batch, far after
    for E in sorted({S + B, S + B // 2 + (1 if B > 1 else 0), S + 1, S, S + B + 7}):
        Eh = hs[E - S].tobytes() if E - S <= B else bytes(32)
        rec, wit = builder.prove_subchain(inp, S, S + B, E, Eh, want_witness=True, raise_on_assert=False)
        ctx = oracle.make_ctx(0, bytes(32), E, Eh)
        rc, ref, cw = oracle.prove_subchain(B, inp["start_header_hash"], inp["end_header_hash"], inp["data_hash_proofs"],
                                            inp["last_block_id_proofs"], S, S + B, E, Eh, ctx=ctx, want_witness=True)
        assert rec_bytes(rec) == rec_bytes(ref), (B, E - S)
        want = oracle.expand_witness(T.map_layout(B), 1, cw)
        assert wit.shape == want.shape
        bad = np.nonzero(wit != want)[0]
        assert bad.size == 0, (B, E - S, bad[:8], wit[bad[:8]], want[bad[:8]])


def test_prove_subchain_detects_every_tamper(builder):
    B = 8
    w = synth.Workload(77, 1, 1, B, v=2)
    S = int(w.first_height[0])
    f = InputDataFetcher(w.headers[0], S, S + B + 2)
    base = f.get_data_commitment_inputs(S, S + B, B)
    Eh = w.hashes[0, B].tobytes()
    cases = []
    t = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
    t["data_hash_proofs"][3]["leaf"][10] ^= 1
    cases.append(t)                                        # A4 at slot 3
    t = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
    t["last_block_id_proofs"][5]["aunts"][2][0] ^= 0x80
    cases.append(t)                                        # lb root of slot 5 changes -> A3 at slot 6 (and A5/A6 later)
    t = dict(base)
    t["start_header_hash"] = bytes(32)
    cases.append(t)                                        # A3 at slot 0
    t = dict(base)
    t["end_header_hash"] = bytes(32)
    cases.append(t)                                        # only matters when the batch ends before E
    for E in (S + B, S + B + 5):
        for c in cases:
            rec, _ = builder.prove_subchain(c, S, S + B, E, Eh, raise_on_assert=False)
            rc, ref, _ = oracle.prove_subchain(B, c["start_header_hash"], c["end_header_hash"], c["data_hash_proofs"],
                                               c["last_block_id_proofs"], S, S + B, E, Eh)
            assert rec_bytes(rec) == rec_bytes(ref)
    rec, _ = builder.prove_subchain(cases[0], S, S + B, S + B, Eh, raise_on_assert=False)
    assert rec["assert_fail"] & T.A4_DATA_HASH_PROOF and rec["first_bad_slot"] == 3
    with pytest.raises(_lib.BsxError) as ei:
        builder.prove_subchain(cases[0], S, S + B, S + B, Eh)
    assert ei.value.status == T.ERR_ASSERT


@pytest.mark.parametrize("J,B,n_blocks", [(2, 32, 64), (2, 32, 35), (8, 32, 256), (8, 32, 131), (16, 16, 1), (32, 32, 1024),
                                          (32, 64, 2048), (32, 64, 1027)])
def test_prove_data_commitment_vs_oracle(J, B, n_blocks, builder):
    """BASELINE configs: 64 = 2x32, 1024 = 32x32, 2048 = 32x64; plus E = S + J*B/2 + 3 (disabled slots and batches)."""
    w = synth.Workload(2, 1, J, B, v=1, n_blocks=n_blocks)
    S = int(w.first_height[0])
    rg = w.ranges[0]
    f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
    out = builder.prove_data_commitment(f, J, B, S, bytes(rg["start_header_hash"]), int(rg["end_block"]), bytes(rg["end_header_hash"]),
                                        want_witness=True)
    rc, ref = oracle.prove_data_commitment(J, B, w.ranges[0:1], w.headers[0], S, int(w.latest[0]), want_witness=True)
    assert rc == T.OK and out["rc"] == T.OK
    assert out["data_commitment"] == ref["data_commitment"]
    assert [rec_bytes(r) for r in out["records"]] == [rec_bytes(r) for r in ref["records"]]
    want = oracle.expand_range_witness(J, B, ref["compact"])
    assert out["witness"].shape == want.shape
    assert (out["witness"] == want).all()
    # size-independent property: the commitment is the RFC 6962 root over the n_blocks tuples
    import hashlib

    def root(items):
        if len(items) == 1:
            return hashlib.sha256(b"\x00" + items[0]).digest()
        k = 1
        while k * 2 < len(items):
            k *= 2
        return hashlib.sha256(b"\x01" + root(items[:k]) + root(items[k:])).digest()
    tuples = [bytes(24) + (S + i).to_bytes(8, "big") + w.headers[0][i]["hash"][1][2:34].tobytes() for i in range(n_blocks)]
    assert out["data_commitment"] == root(tuples)


def test_prove_data_commitment_broken_chain_matches_oracle(builder):
    J, B = 4, 8
    w = synth.Workload(3, 1, J, B, v=1)
    S = int(w.first_height[0])
    hdrs = w.headers[0].copy()
    hdrs[13]["hash"][1][7] ^= 0x55          # data hash of header 13 changes -> its hash changes -> link 13->14 breaks
    rg = w.ranges[0]
    f = InputDataFetcher(hdrs, S, int(w.latest[0]))
    out = builder.prove_data_commitment(f, J, B, S, bytes(rg["start_header_hash"]), int(rg["end_block"]), bytes(rg["end_header_hash"]),
                                        raise_on_assert=False)
    rc, ref = oracle.prove_data_commitment(J, B, w.ranges[0:1], hdrs, S, int(w.latest[0]))
    assert out["rc"] == rc == T.ERR_ASSERT
    assert out["result"]["assert_fail"] == ref["status"] != 0
    assert [rec_bytes(r) for r in out["records"]] == [rec_bytes(r) for r in ref["records"]]


@pytest.mark.parametrize("v,v_max,absent,nc", [(100, 100, 0, 3), (100, 128, 150, 3), (7, 8, 0, 3), (512, 512, 300, 3), (1, 1, 0, 3),
                                                (20, 24, 100, 9), (100, 100, 50, 12)])   # nc >= 8: fixed-key P7 inside the host tier
def test_verify_commits_vs_oracle(v, v_max, absent, nc):
    w = synth.Workload(40 + v, nc, 1, 2, v=v, v_max=v_max, absent_permille=absent)
    vals = w.validators.copy()                      # [3, v_max]
    hh = w.commit_hashes.copy()
    rnd = np.random.default_rng(v)
    # tamper: flip a signature bit, a message byte inside the hash, a pubkey bit; make s non-canonical
    for c in range(nc):
        signed = np.nonzero(vals[c]["is_signed"])[0]
        if signed.size >= 4:
            a, b, d, e = signed[rnd.permutation(signed.size)[:4]]
            vals[c, a]["signature"][int(rnd.integers(0, 64))] ^= 1 << int(rnd.integers(0, 8))
            vals[c, b]["message"][20] ^= 1
            vals[c, d]["pubkey"][int(rnd.integers(0, 32))] ^= 1 << int(rnd.integers(0, 8))
            s = int.from_bytes(bytes(vals[c, e]["signature"][32:]), "little") + (2 ** 252 + 27742317777372353535851937790883648493)
            vals[c, e]["signature"][32:] = np.frombuffer((s % 2 ** 256).to_bytes(32, "little"), np.uint8)
    res, ok = verify_commits(vals, hh)
    for c in range(nc):
        ref, rok = oracle.verify_commit(vals[c], hh[c].tobytes())
        assert (ok[c] == rok).all(), (c, np.nonzero(ok[c] != rok))
        assert res_bytes(res[c]) == res_bytes(ref), c


@pytest.mark.parametrize("n_keys", [24, 10, 0])
def test_ed25519_fixed_key_path_matches_generic_and_oracle(n_keys):
    """bsx_dev_ed25519_verify_keyed (per-validator tables) against bsx_dev_ed25519_verify and the oracle, with slots
    that must take the table path, slots that must fall back (key differs from the table row, row missing) and a
    table row whose key does not decode."""
    import ctypes as C
    import torch
    n_commits, v, v_max = 6, 22, 24
    w = synth.Workload(77, n_commits, 1, 2, v=v, v_max=v_max, absent_permille=100)
    vals = w.validators.copy()
    hh = w.commit_hashes.copy()
    L_ = 2 ** 252 + 27742317777372353535851937790883648493
    vals[1, 1]["signature"][7] ^= 4                      # bad R, table path
    vals[1, 3]["signature"][40] ^= 1                     # bad s, table path
    vals[2, 4]["pubkey"][9] ^= 2                         # key differs from row 4 -> fallback, and it does not verify
    vals[3, 5] = vals[3, 6]                              # another validator's valid record in slot 5 -> fallback, verifies
    vals[3, 5]["enabled"] = 1
    s = int.from_bytes(bytes(vals[4, 8]["signature"][32:]), "little") + L_
    vals[4, 8]["signature"][32:] = np.frombuffer((s % 2 ** 256).to_bytes(32, "little"), np.uint8)   # s + L
    bad_key = np.frombuffer((2).to_bytes(32, "little"), np.uint8)    # y = 2 is not on the curve
    for c in range(n_commits):
        vals[c, 7]["pubkey"] = bad_key                   # table row 7 does not decode; every commit carries the same key
    vals[5, 9]["message"][30] ^= 1                       # challenge changes, table path
    flat = vals.reshape(-1)
    n = flat.size
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    dv = torch.from_numpy(flat.view(np.uint8).copy()).cuda()
    dh = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    ok_g = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ok_k = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    tab = torch.zeros(max(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(n_keys))), 16), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
    _lib.check(L.bsx_dev_ed25519_verify(ctx, st, dp(dv), dp(dh), C.c_uint64(n), dp(ok_g)))
    _lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(n_keys), dp(tab)))
    _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(v_max), dp(tab),
                                              C.c_uint32(n_keys), dp(ok_k), None))
    # the same with the batch-inversion scratch (projective results parked, 8 encodings per inversion): identical verdicts
    ok_b = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    scr = torch.zeros(int(L.bsx_ed25519_verify_scratch_bytes(C.c_uint64(n))), dtype=torch.uint8, device="cuda")
    _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(v_max), dp(tab),
                                              C.c_uint32(n_keys), dp(ok_b), dp(scr)))
    torch.cuda.synchronize()
    assert torch.equal(ok_b, ok_k)
    ok_g, ok_k = ok_g.cpu().numpy().reshape(n_commits, v_max), ok_k.cpu().numpy().reshape(n_commits, v_max)
    for c in range(n_commits):
        _, rok = oracle.verify_commit(vals[c], hh[c].tobytes())
        assert (ok_k[c] == rok).all(), (c, np.nonzero(ok_k[c] != rok))
        assert (ok_g[c] == rok).all(), c
    assert ok_k[3, 5] == 1 and ok_k[2, 4] == 0 and ok_k[1, 1] == 0 and ok_k[1, 3] == 0 and ok_k[4, 8] == 0
    assert (ok_k[:, 7] == 0).all() and ok_k[5, 9] == 0
    assert ok_k.sum() > n_commits * v // 2


def test_ed25519_key_table_reuse_across_calls_and_validator_set_changes():
    """bsx_dev_ed25519_keytable keeps rows whose key is unchanged (one compare per row) and rebuilds exactly the rows that
    changed: the same table buffer is driven through set A, set A again (pure reuse), set A with three slots re-keyed, a
    different n_keys (layout change) and back; after every call the keyed verdicts must equal the oracle's, and the
    stale-row hazard (a reused row must never serve another key) is probed by swapping two validators' slots."""
    import ctypes as C
    import torch
    v = 20
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(v))), dtype=torch.uint8, device="cuda")
    wA = synth.Workload(70, 3, 1, 2, v=v)
    wB = synth.Workload(71, 3, 1, 2, v=v)              # a different validator set

    def run(vals, hashes, n_keys):
        flat = np.ascontiguousarray(vals).reshape(-1)
        n = flat.size
        dv = torch.from_numpy(flat.view(np.uint8).copy()).cuda()
        dh = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
        _lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(n_keys), dp(tab)))
        scr = torch.zeros(int(L.bsx_ed25519_verify_scratch_bytes(C.c_uint64(n))), dtype=torch.uint8, device="cuda")
        _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(v), dp(tab), C.c_uint32(n_keys), dp(ok),
                                                  dp(scr) if vals.shape[0] % 2 else None))
        torch.cuda.synchronize()
        ok = ok.cpu().numpy().reshape(vals.shape[0], v)
        for c in range(vals.shape[0]):
            _, rok = oracle.verify_commit(vals[c], hashes[c].tobytes())
            assert (ok[c] == rok).all(), c
        dirty = tab[:n_keys * 64].cpu().numpy().view(np.uint32).reshape(n_keys, 16)[:, 14]
        return ok, dirty

    ok, dirty = run(wA.validators, wA.commit_hashes, v)
    assert ok.all() and dirty.all()                                   # cold: every row built
    ok, dirty = run(wA.validators, wA.commit_hashes, v)
    assert ok.all() and not dirty.any()                               # warm: nothing rebuilt, same verdicts
    mixed = wA.validators.copy()
    for slot in (2, 7, 19):
        mixed[:, slot] = wB.validators[:, slot]                       # three validators replaced (their own valid signatures
    hh = wA.commit_hashes                                             # are over wB's hashes: message check is the tally's job)
    ok, dirty = run(mixed, hh, v)
    assert list(np.nonzero(dirty)[0]) == [2, 7, 19] and ok.all()
    swapped = wA.validators.copy()
    swapped[:, [4, 5]] = swapped[:, [5, 4]]                            # same keys, other rows: both rows must be rebuilt
    ok, dirty = run(swapped, wA.commit_hashes, v)
    assert set(np.nonzero(dirty)[0]) == {2, 4, 5, 7, 19} and ok.all()
    ok, dirty = run(wA.validators[:, :], wA.commit_hashes, 12)         # fewer table rows: layout changes, slots >= 12 fall back
    assert dirty.all() and ok.all()
    ok, dirty = run(wA.validators, wA.commit_hashes, v)
    assert dirty.all() and ok.all()
    bad = wA.validators.copy()
    bad[1, 3]["signature"][3] ^= 1
    ok, dirty = run(bad, wA.commit_hashes, v)
    assert not dirty.any() and ok[1, 3] == 0 and ok.sum() == ok.size - 1


def test_sha512_challenge_vs_oracle():
    import ctypes as C
    import torch
    w = synth.Workload(50, 2, 1, 2, v=64)
    vals = w.validators.reshape(-1).copy()
    vals[5]["message_len"] = 0
    vals[6]["message_len"] = 47
    vals[7]["message_len"] = 48
    vals[8]["message_len"] = 124
    vals[8]["message"][:] = 0xAB
    n = vals.size
    dv = torch.from_numpy(vals.view(np.uint8).copy()).cuda()
    dh = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    dd = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().bsx_dev_sha512_challenge(_lib.context(0), st, _lib.dp(dv), C.c_uint64(n), _lib.dp(dh), _lib.dp(dd)))
    torch.cuda.synchronize()
    h, dig = oracle.sha512_challenge(vals)
    assert (dh.cpu().numpy().reshape(n, 32) == h).all()
    assert (dd.cpu().numpy().reshape(n, 64) == dig).all()


@pytest.mark.parametrize("J,B,v,n_blocks", [(2, 32, 100, 64), (2, 32, 100, 35), (32, 32, 100, 1024), (32, 64, 100, 2048)])
def test_header_range_vs_oracle(J, B, v, n_blocks):
    """BASELINE configs #2-#4 on one GPU, mode F (one commit per range), full witness diff."""
    w = synth.Workload(1, 1, J, B, v=v, n_blocks=n_blocks)
    S = int(w.first_height[0])
    f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
    circ = CombinedSkipCircuit(v, J, B)
    out, res, wit = circ.prove(w.input48(0), f, w.validators[0], w.trusted[0], want_witness=True)
    rc, ref_out, ref_res, cw = oracle.header_range(J, B, w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0],
                                                   want_witness=True)
    assert rc == T.OK
    assert out == ref_out and out[:32] == w.hashes[0, n_blocks].tobytes()
    assert res_bytes(res) == res_bytes(ref_res)
    # the WHOLE circuit's witness: map jobs, reduce nodes, the COMMIT unit of the target commit, the SKIP unit (round 4)
    assert wit.size == T.header_range_witness_elements(J, B, v)
    assert (wit == oracle.expand_range_witness(J, B, cw, v_max=v)).all()


def test_chain_id_is_checked_on_gpu_like_the_oracle():
    """a header of another chain signed by the same validator keys must not get BSX_OK (builder.skip / step are
    called with C::CHAIN_ID_BYTES, header_range.rs:42-43, next_header.rs:32-33) — host tier, step circuit and the batched
    engine, every verdict equal to the oracle's."""
    from blobstreamx_amd.engine import HeaderRangeEngine
    for fork_id in ("celestiaX", "celestib", "mocha-4", "c"):
        w = synth.Workload(98, 1, 2, 4, v=4, chain_id=fork_id)
        S = int(w.first_height[0])
        f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
        for cid in (fork_id.encode(), b"celestia"):
            want = oracle.header_range(2, 4, w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0], chain_id=cid)[0]
            assert want == (T.OK if cid == fork_id.encode() else T.ERR_ASSERT)
            try:
                CombinedSkipCircuit(4, 2, 4, chain_id=cid).prove(w.input48(0), f, w.validators[0], w.trusted[0])
                rc = T.OK
            except _lib.BsxError as e:
                rc = e.status
            assert rc == want, (fork_id, cid, rc)
            eng = HeaderRangeEngine(2, 4, 4, 1, chain_id=cid)
            eng.upload_workload(w)
            eng.step()
            assert eng.download()["skip_status"][0] == want
        wS = synth.Workload(98, 1, 2, 4, v=4, chain_id=fork_id, mode="S")
        inp = int(wS.first_height[0]).to_bytes(8, "big") + wS.hashes[0, 0].tobytes()
        vals = wS.validators[0].reshape(-1, 4)[0]
        for cid in (fork_id.encode(), b"celestia"):
            want = oracle.next_header(inp, wS.headers[0, 0], wS.headers[0, 1], int(wS.latest[0]), vals, chain_id=cid)[0]
            try:
                CombinedStepCircuit(4, chain_id=cid).prove(inp, wS.headers[0, 0], wS.headers[0, 1], int(wS.latest[0]), vals)
                rc = T.OK
            except _lib.BsxError as e:
                rc = e.status
            assert rc == want == (T.OK if cid == fork_id.encode() else T.ERR_ASSERT)


@pytest.mark.parametrize("rnd,nil,absent", [(3, 250, 150), (1 << 40, 0, 0), (0, 400, 0), (9, 0, 600)])
def test_commit_round_nonzero_nil_and_absent_votes_on_gpu(rnd, nil, absent):
    """round != 0 sign-bytes (block hash at offset 25), BlockIDFlag Nil and Absent validators — per
    signature ok bits, commit results and the header_range verdict (which may be BSX_ERR_VOTING_POWER when too few signed)
    equal the oracle's, through both Ed25519 paths (8 commits -> fixed-key tables, 1 commit -> generic)."""
    V = 24
    w = synth.Workload(35, 8, 2, 4, v=V, round=rnd, nil_permille=nil, absent_permille=absent)
    res, ok = verify_commits(w.validators, w.commit_hashes)
    for c in range(8):
        want, wok = oracle.verify_commit(w.validators[c], w.commit_hashes[c].tobytes())
        assert (ok[c] == wok).all() and res_bytes(res[c]) == res_bytes(want), c
        signed = w.validators[c]["is_signed"] != 0
        assert (ok[c] == signed).all() and want["n_bad_message"] == 0
    res1, ok1 = verify_commits(w.validators[:1], w.commit_hashes[:1])
    assert (ok1[0] == ok[0]).all() and res_bytes(res1[0]) == res_bytes(res[0])
    bad = w.validators[:1].copy()
    k = int(np.nonzero(bad[0]["is_signed"])[0][0])
    off = 25 if rnd else 16
    bad[0, k]["message"][off + 3] ^= 0x10                         # the carried block hash no longer matches
    r2, o2 = verify_commits(bad, w.commit_hashes[:1])
    want, wok = oracle.verify_commit(bad[0], w.commit_hashes[0].tobytes())
    assert res_bytes(r2[0]) == res_bytes(want) and (o2[0] == wok).all() and want["n_bad_message"] == 1
    for r in range(3):
        S = int(w.first_height[r])
        rc = oracle.header_range(2, 4, w.input48(r), w.headers[r], S, int(w.latest[r]), w.validators[r], w.trusted[r])[0]
        try:
            CombinedSkipCircuit(V, 2, 4).prove(w.input48(r), InputDataFetcher(w.headers[r], S, int(w.latest[r])), w.validators[r], w.trusted[r])
            got = T.OK
        except _lib.BsxError as e:
            got = e.status
        assert got == rc, (r, got, rc)
        assert rc in (T.OK, T.ERR_VOTING_POWER)


def test_mode_s_full_size_2048_x_512_vs_oracle():
    """BASELINE config #5 in mode S at FULL size on one GPU: one header_range_2048 whose every header carries its own
    512-validator commit = 1,048,576 signatures (next_header.rs:25-47 per header).  Every per-signature ok bit and all 2048
    commit results (validators hash, tallies, 2/3 flag, message checks) against the oracle run on all host threads; a
    sprinkle of tampered signatures / messages / absent validators keeps the verdicts from being all-ones."""
    import os
    J, B, V = 32, 64, 512
    w = synth.Workload(5, 1, J, B, v=V, mode="S", absent_permille=30)
    vals = w.validators.reshape(J * B, V).copy()
    rng = np.random.default_rng(5)
    for c in rng.integers(0, J * B, 40):
        k = int(rng.integers(0, V))
        which = int(rng.integers(0, 3))
        if which == 0:
            vals[c, k]["signature"][int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        elif which == 1:
            vals[c, k]["message"][20] ^= 4                           # inside the block hash: bad message AND bad signature
        else:
            vals[c, k]["pubkey"][int(rng.integers(0, 31))] ^= 2       # key no longer matches the table row -> generic fallback
    res, ok = verify_commits(vals, w.commit_hashes)
    want, wok = oracle.bench_verify_commits(vals, w.commit_hashes, min(32, os.cpu_count() or 1))     # the GPU boxes grant ~16 CPUs
    assert (ok == wok).all(), np.argwhere(ok != wok)[:5]
    a, b = res.copy(), want.copy()
    a["_pad"] = 0; b["_pad"] = 0
    assert a.tobytes() == b.tobytes()
    assert 0 < (ok == 0).sum() < 0.05 * ok.size and res["n_bad_signature"].sum() >= 20 and res["two_thirds_ok"].all()


def test_witness_manifest_decodes_a_real_witness():
    """bsx_witness_manifest used the way a plonky2x shim would: slice the expanded witness of every map job by variable
    name and check the decoded VALUES against what the ABI returns separately (records, hint output, public output)."""
    from blobstreamx_amd.builder import witness_manifest, witness_view
    J, B, V = 4, 8, 6
    w = synth.Workload(44, 1, J, B, v=V, n_blocks=27)
    S = int(w.first_height[0])
    f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
    out, _, wit = CombinedSkipCircuit(V, J, B).prove(w.input48(0), f, w.validators[0], w.trusted[0], want_witness=True)
    bld = DataCommitmentBuilder()
    res = bld.prove_data_commitment(f, J, B, S, w.hashes[0, 0].tobytes(), S + 27, w.hashes[0, 27].tobytes())
    m, mr = witness_manifest(B), witness_manifest(0)
    nel, rel = int(T.map_layout(B)["n_elements"]), int(T.reduce_layout()["n_elements"])
    pack = lambda v: np.packbits(v.astype(np.uint8), axis=1)
    u64 = lambda v: int(v[0, 0]) | (int(v[0, 1]) << 32)
    for j in range(J):
        wj = wit[j * nel:(j + 1) * nel]
        rec = res["records"][j]
        assert pack(witness_view(m, wj, "record.data_merkle_root"))[0].tobytes() == bytes(rec["data_merkle_root"])
        assert pack(witness_view(m, wj, "record.end_header"))[0].tobytes() == bytes(rec["end_header"])
        assert u64(witness_view(m, wj, "record.start_block")) == int(rec["start_block"]) == S + j * B
        assert u64(witness_view(m, wj, "record.end_block")) == int(rec["end_block"])
        assert int(witness_view(m, wj, "record.is_enabled")[0, 0]) == int(rec["is_enabled"])
        assert u64(witness_view(m, wj, "batch_end_block")) == S + (j + 1) * B
        inp = f.get_data_commitment_inputs(S + j * B, S + (j + 1) * B, B)
        assert (pack(witness_view(m, wj, "data_comm_proof.data_hash_proofs[].leaf")) == inp["data_hash_proofs"]["leaf"]).all()
        assert (pack(witness_view(m, wj, "data_comm_proof.last_block_id_proofs[].proof")).reshape(B, 4, 32) == inp["last_block_id_proofs"]["aunts"]).all()
        heights = witness_view(m, wj, "block_height[]")
        assert [int(h[0]) | (int(h[1]) << 32) for h in heights] == [S + j * B + i for i in range(B)]
        roots = pack(witness_view(m, wj, "slot[].data_hash_path"))[:, 128:160]        # last of the 5 path digests
        for i in range(B):
            if S + j * B + i < S + 27:
                assert roots[i].tobytes() == w.hashes[0, j * B + i].tobytes()          # data_hash_proof_root == header hash
    top = wit[J * nel + (J - 2) * rel:]                                                # the last reduce node = the range result
    assert pack(witness_view(mr, top, "out.data_merkle_root"))[0].tobytes() == out[32:]
    with pytest.raises(KeyError):
        witness_view(m, wit[:nel], "no_such_variable")


def test_voting_power_overflow_guard_on_gpu():
    """totals beyond MaxTotalVotingPower are flagged by the tally / skip-eval kernels and rejected with
    BSX_ERR_BAD_ARG by every entry point that consumes them, exactly as the oracle does."""
    from blobstreamx_amd.builder import find_block_to_request
    w = synth.Workload(36, 1, 2, 4, v=6)
    S = int(w.first_height[0])
    f = InputDataFetcher(w.headers[0], S, int(w.latest[0]))
    big = w.validators[0].copy()
    big["voting_power"][:4] = (1 << 62)
    big["is_signed"][:4] = 0
    with pytest.raises(_lib.BsxError) as ei:
        verify_commits(big[None], w.commit_hashes[:1])
    assert ei.value.status == T.ERR_BAD_ARG
    edge = w.validators[0].copy()
    edge["voting_power"] = 0
    edge["voting_power"][0] = 1152921504606846975
    res, _ = verify_commits(edge[None], w.commit_hashes[:1])
    want, _ = oracle.verify_commit(edge, w.commit_hashes[0].tobytes())
    assert res_bytes(res[0]) == res_bytes(want) and res[0]["power_overflow"] == 0
    tr = w.trusted[0].copy()
    tr["voting_power"][:] = (1 << 61)
    for tv, rv in ((big, w.trusted[0]), (w.validators[0], tr)):
        assert oracle.header_range(2, 4, w.input48(0), w.headers[0], S, int(w.latest[0]), tv, rv)[0] == T.ERR_BAD_ARG
        with pytest.raises(_lib.BsxError) as ei:
            CombinedSkipCircuit(6, 2, 4).prove(w.input48(0), f, tv, rv)
        assert ei.value.status == T.ERR_BAD_ARG
    s2 = w.trusted[0].copy()
    s2["voting_power"][:] = (1 << 61)
    with pytest.raises(_lib.BsxError) as ei:
        find_block_to_request(1000, 1002, s2, [1002], w.validators[:1])
    assert ei.value.status == T.ERR_BAD_ARG


def test_header_range_failure_codes_match_oracle():
    J, B, v = 2, 4, 10
    circ = CombinedSkipCircuit(v, J, B)

    def both(w, vals=None, trusted=None, inp=None):
        vals = w.validators[0] if vals is None else vals
        trusted = w.trusted[0] if trusted is None else trusted
        inp = w.input48(0) if inp is None else inp
        S = int(w.first_height[0])
        rc_ref = oracle.header_range(J, B, inp, w.headers[0], S, int(w.latest[0]), vals, trusted)[0]
        try:
            circ.prove(inp, InputDataFetcher(w.headers[0], S, int(w.latest[0])), vals, trusted)
            rc = T.OK
        except _lib.BsxError as e:
            rc = e.status
        assert rc == rc_ref, (rc, rc_ref)
        return rc

    w = synth.Workload(60, 1, J, B, v=v)
    assert both(w) == T.OK
    bad = w.validators[0].copy()
    bad[2]["signature"][1] ^= 4
    assert both(w, vals=bad) == T.ERR_BAD_SIGNATURE
    w2 = synth.Workload(61, 1, J, B, v=v, absent_permille=600)      # < 2/3 signed
    assert both(w2) == T.ERR_VOTING_POWER
    tr = w.trusted[0].copy()
    tr[0]["voting_power"] += 1                                        # trusted set no longer hashes to the header's field 7
    assert both(w, trusted=tr) == T.ERR_ASSERT
    inp = bytearray(w.input48(0))
    inp[10] ^= 1                                                      # wrong trusted header hash
    assert both(w, inp=bytes(inp)) == T.ERR_ASSERT
    inp = w.input48(0)[:40] + (int(w.ranges[0]["start_block"]) + J * B + 1).to_bytes(8, "big")
    assert both(w, inp=inp) == T.ERR_RANGE_TOO_LONG


def test_expand_witness_property_full_size():
    """size-independent property at the 2048 shape: packing the bit elements back gives the compact bytes, i.e. the
    expansion kernel is the exact inverse of np.packbits over every byte of every map job."""
    import ctypes as C
    import torch
    B, J = 64, 32
    lay = T.map_layout(B)
    stride, n_el = int(lay["compact_stride"]), int(lay["n_elements"])
    rnd = np.random.default_rng(1)
    compact = rnd.integers(0, 256, size=J * stride, dtype=np.uint8)
    dc = torch.from_numpy(compact).cuda()
    dw = torch.zeros(J * n_el, dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    layout = np.array(lay).reshape(1)
    _lib.check(_lib.lib().bsx_dev_expand_witness(_lib.context(0), st, _lib.p(layout), C.c_uint32(J), _lib.dp(dc), _lib.dp(dw)))
    torch.cuda.synchronize()
    got = dw.cpu().numpy().view(np.uint64)
    want = oracle.expand_witness(lay, J, compact)
    assert (got == want).all()


# ------------------------------------------------------------------ §8(f) row 3: operator skip-target search
@pytest.mark.parametrize("seed,v,v_max", [(1, 9, 9), (2, 100, 100), (3, 37, 64), (4, 512, 512)])
def test_find_block_to_request_vs_oracle(seed, v, v_max):
    """bsx_find_block_to_request (one launch evaluating every candidate of the halving sequence, then the reference's
    loop, fetcher.rs:60-87) against the oracle: block chosen and every per-candidate tally.  Candidates drift away from
    the start set the further they are (validators replaced, signers dropping out), so the search really descends."""
    from blobstreamx_amd.builder import find_block_to_request, halving_sequence
    rnd = np.random.default_rng(seed)
    start = synth.ValidatorSet(100 + seed, v).as_validators(v_max)
    S = 5000
    for M in (S + 1, S + 2, S + 777, S + 2048):
        heights = halving_sequence(S, M)
        cands = []
        for h in heights:
            c = start.copy()
            far = (h - S) / 2048.0                                  # 0 near the start block, 1 at the far end
            for i in range(v):
                if rnd.random() < 0.9 * far:                        # replaced by a stranger
                    c[i]["pubkey"] = rnd.integers(0, 256, 32, dtype=np.uint8)
                c[i]["is_signed"] = 1 if rnd.random() < 0.8 else 0
                if rnd.random() < 0.1:
                    c[i]["voting_power"] = int(rnd.integers(1, 10 ** 9))   # the candidate's own power, never used for the overlap
            cands.append(c)
        cands = np.stack(cands) if cands else np.zeros((0, v_max), T.VALIDATOR)
        rc, want, wev = oracle.find_block_to_request(S, M, start, heights, cands)
        blk, ev = find_block_to_request(S, M, start, heights, cands)
        assert rc == T.OK and blk == want, (M, blk, want)
        for f in ("overlap_power", "start_total_power", "signed_power", "target_total_power", "valid"):
            assert (ev[f] == wev[f]).all(), (M, f)
    with pytest.raises(_lib.BsxError) as ei:                       # a visited height is missing
        find_block_to_request(S, S + 100, start, [S + 100], np.stack([np.zeros(v_max, T.VALIDATOR)]))
    assert ei.value.status == T.ERR_BAD_ARG


# ------------------------------------------------------------------ next_header (CombinedStepCircuit)
def test_golden_next_header_circuit(golden, mocha):
    """bsx_next_header over the fixture chain: next header hash ‖ data commitment, equal to the oracle and to the
    reference's fixture commitments; reject paths return the oracle's codes."""
    from blobstreamx_amd.builder import CombinedStepCircuit
    circ = CombinedStepCircuit(4, chain_id=b"mocha-4")
    for k in range(4):
        h = 10000 + k
        inp = h.to_bytes(8, "big") + mocha["hashes"][k]
        out, cr = circ.prove(inp, mocha["headers"][k], mocha["headers"][k + 1], mocha["latest"], mocha["commits"][k + 1])
        rc, want, wcr = oracle.next_header(inp, mocha["headers"][k], mocha["headers"][k + 1], mocha["latest"], mocha["commits"][k + 1], chain_id=b"mocha-4")
        assert rc == T.OK and out == want and res_bytes(cr) == res_bytes(wcr)
        assert out[:32] == mocha["hashes"][k + 1]
        if f"{h}-{h + 1}" in golden["data_commitments"]:
            assert out[32:].hex() == golden["data_commitments"][f"{h}-{h + 1}"]
    inp = (10000).to_bytes(8, "big") + mocha["hashes"][0]
    cases = [((10000).to_bytes(8, "big") + mocha["hashes"][1], 0, 1, 1), (inp, 0, 1, 2), (inp, 0, 2, 2), ((10001).to_bytes(8, "big") + mocha["hashes"][0], 0, 1, 1)]
    for i, a, b, c in cases:
        rc = oracle.next_header(i, mocha["headers"][a], mocha["headers"][b], mocha["latest"], mocha["commits"][c], chain_id=b"mocha-4")[0]
        assert rc != T.OK
        with pytest.raises(_lib.BsxError) as ei:
            circ.prove(i, mocha["headers"][a], mocha["headers"][b], mocha["latest"], mocha["commits"][c])
        assert ei.value.status == rc, (a, b, c)


@pytest.mark.parametrize("v,v_max", [(100, 100), (7, 8), (33, 64)])
def test_next_header_vs_oracle(v, v_max):
    """Synthetic chains (mode S: a commit on every header): every step of a few ranges, plus tampering of each link
    the step enforces."""
    from blobstreamx_amd.builder import CombinedStepCircuit
    w = synth.Workload(60 + v, 2, 1, 8, v=v, v_max=v_max, mode="S", absent_permille=100)
    circ = CombinedStepCircuit(v_max)
    per = w.hpr - 1
    n_ok = 0
    for r in range(2):
        S = int(w.first_height[r])
        for k in range(per):
            inp = (S + k).to_bytes(8, "big") + w.hashes[r, k].tobytes()
            vals = w.validators[r * per + k]
            rc, want, wcr = oracle.next_header(inp, w.headers[r, k], w.headers[r, k + 1], int(w.latest[r]), vals)
            if rc == T.OK:
                out, cr = circ.prove(inp, w.headers[r, k], w.headers[r, k + 1], int(w.latest[r]), vals)
                assert out == want and res_bytes(cr) == res_bytes(wcr), (r, k)
                assert out[:32] == w.hashes[r, k + 1].tobytes()
                n_ok += 1
            else:                                   # a small set with absent signers can miss 2/3: same verdict required
                with pytest.raises(_lib.BsxError) as ei:
                    circ.prove(inp, w.headers[r, k], w.headers[r, k + 1], int(w.latest[r]), vals)
                assert ei.value.status == rc == T.ERR_VOTING_POWER, (r, k, rc)
    assert n_ok >= per
    r, k = 0, 2
    S = int(w.first_height[r])
    inp = (S + k).to_bytes(8, "big") + w.hashes[r, k].tobytes()
    vals = w.validators[r * per + k]

    def both(i, ph, nh, vv):
        rc = oracle.next_header(i, ph, nh, int(w.latest[r]), vv)[0]
        try:
            circ.prove(i, ph, nh, int(w.latest[r]), vv)
            got = T.OK
        except _lib.BsxError as e:
            got = e.status
        assert got == rc, (got, rc)
        return rc
    nh = w.headers[r, k + 1].copy(); nh["last_block_id"][5] ^= 1
    assert both(inp, w.headers[r, k], nh, vals) != T.OK                       # header changed: commit no longer signs it
    ph = w.headers[r, k].copy(); ph["hash"][3][9] ^= 1
    assert both(inp, ph, w.headers[r, k + 1], vals) == T.ERR_ASSERT           # prev no longer hashes to the public input
    vv = vals.copy(); s = np.nonzero(vv["is_signed"])[0][0]; vv[s]["signature"][3] ^= 8
    assert both(inp, w.headers[r, k], w.headers[r, k + 1], vv) == T.ERR_BAD_SIGNATURE
    vv = vals.copy(); vv["is_signed"][: (2 * v) // 3] = 0
    assert both(inp, w.headers[r, k], w.headers[r, k + 1], vv) == T.ERR_VOTING_POWER
    vv = vals.copy(); vv[0]["voting_power"] += 1
    assert both(inp, w.headers[r, k], w.headers[r, k + 1], vv) == T.ERR_ASSERT   # validator set no longer hashes to validators_hash
    assert both(inp, w.headers[r, k], w.headers[r, k + 2], w.validators[r * per + k + 1]) == T.ERR_ASSERT   # skipping a block is not a step


def test_tuning_and_launch_forms_do_not_change_results():
    """The launch-form knobs of bsx_pipeline_config (tune_merkle_workgroups: grid-strided header hashing; tune_subchain:
    prove_subchain as one launch or as separate launches; BSX_PIPE_RECOMPUTE_PATHS) and bsx_set_tuning for the device tier are
    performance knobs: same hashes, proofs, path digests, records and compact witness."""
    import ctypes as C
    from blobstreamx_amd.engine import HeaderRangeEngine
    J, B, V, R = 8, 32, 10, 3
    w = synth.Workload(21, R, J, B, v=V, n_blocks=J * B - 5)
    L = _lib.lib()
    outs = []
    for wgs, form, fused in ((0, 1, True), (3, 1, True), (512, 2, True), (0xffffffff, 0, False)):
        eng = HeaderRangeEngine(J, B, V, R, with_witness=False, merkle_workgroups=wgs, subchain_form=form, fused_hint=fused)
        eng.upload_workload(w)
        eng.step()
        res = eng.download()
        outs.append((res["output64"].tobytes(), res["records"].tobytes(), eng.hashes_all.cpu().numpy().tobytes(),
                     eng.dh_aunts.cpu().numpy().tobytes(), eng.paths.cpu().numpy().tobytes() if eng.paths is not None else b"",
                     eng.compact.cpu().numpy().tobytes()))
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][:4] == outs[3][:4] and outs[0][5] == outs[3][5]          # paths recomputed from the proofs: same witness
    assert L.bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_MERKLE_WORKGROUPS), C.c_uint64(7)) == T.OK
    assert L.bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_MERKLE_WORKGROUPS), C.c_uint64(0)) == T.OK
    assert L.bsx_set_tuning(_lib.context(0), C.c_uint32(99), C.c_uint64(1)) == T.ERR_BAD_ARG



def test_status_priority_with_several_faults_matches_the_oracle():
    """an input with several faults at once must get the SAME status from the product and its checker.  Order
    (k_skip_check = orc_header_range): voting-power overflow -> trusted hash -> height leaf -> chain-id leaf -> signatures ->
    validator-set hashes -> 2/3 -> 1/3.  Combined cases: overflow + wrong chain id, overflow + wrong trusted hash,
    wrong chain id + bad signature, bad signature + broken validator-set hash."""
    from blobstreamx_amd.engine import HeaderRangeEngine
    J, B, V = 2, 4, 6

    def run(w, cid=b"celestia", inp=None):
        S = int(w.first_height[0])
        inp = inp if inp is not None else w.input48(0)
        want = oracle.header_range(J, B, inp, w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0], chain_id=cid)[0]
        try:
            CombinedSkipCircuit(V, J, B, chain_id=cid).prove(inp, InputDataFetcher(w.headers[0], S, int(w.latest[0])), w.validators[0], w.trusted[0])
            rc = T.OK
        except _lib.BsxError as e:
            rc = e.status
        assert rc == want, (rc, want)
        eng = HeaderRangeEngine(J, B, V, 1, chain_id=cid)
        eng.upload_workload(w)
        eng.step()
        assert eng.download()["skip_status"][0] == want
        return want

    w = synth.Workload(36, 1, J, B, v=V)
    w.validators[0]["voting_power"][:4] = 1 << 62                       # overflow ...
    assert run(w, cid=b"mocha-4") == T.ERR_BAD_ARG                      # ... + wrong chain id
    bad_in = bytearray(w.input48(0)); bad_in[9] ^= 1
    assert run(w, inp=bytes(bad_in)) == T.ERR_BAD_ARG                   # ... + wrong trusted hash
    w = synth.Workload(36, 1, J, B, v=V)
    w.validators[0, 1]["signature"][2] ^= 1
    assert run(w, cid=b"mocha-4") == T.ERR_ASSERT                       # wrong chain id outranks the bad signature
    w.trusted[0, 0]["voting_power"] += 1
    assert run(w) == T.ERR_BAD_SIGNATURE                                # bad signature outranks the trusted-set hash


def test_header_range_graph_replay_vs_oracle():
    """bsx_header_range replays a captured hipGraph from the fourth request of a shape on (the request's bytes travel through
    the staging block and the header buffer).  Twelve requests over different inputs of one shape — valid ranges, a broken chain,
    a bad signature — then a different range length (new key: direct launches, new capture) and back, and the same sequence
    with graphs disabled: every public output, status and commit result must be the oracle's."""
    import ctypes as C
    J, B, V, R = 4, 16, 10, 6
    w = synth.Workload(77, R, J, B, v=V)
    w.headers[2, 9]["hash"][1][5] ^= 1
    w.validators[4, 3]["signature"][0] ^= 2
    w2 = synth.Workload(78, 2, J, B, v=V, n_blocks=J * B - 7)
    L = _lib.lib()

    def one(ws, r):
        S = int(ws.first_height[r])
        want_rc, want_out, want_res, _ = oracle.header_range(J, B, ws.input48(r), ws.headers[r], S, int(ws.latest[r]), ws.validators[r], ws.trusted[r])
        circ = CombinedSkipCircuit(V, J, B)
        try:
            out, res, _ = circ.prove(ws.input48(r), InputDataFetcher(ws.headers[r], S, int(ws.latest[r])), ws.validators[r], ws.trusted[r])
            rc = T.OK
            assert out == want_out, r
            a, b = np.array(res).copy(), np.array(want_res).copy()
            a["_pad"] = 0; b["_pad"] = 0
            assert a.tobytes() == b.tobytes(), r
        except _lib.BsxError as e:
            rc = e.status
        assert rc == want_rc, (r, rc, want_rc)
        return rc

    for graphs in (1, 0, 1):
        _lib.check(L.bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(graphs)))
        seen = []
        for i in range(12):
            seen.append(one(w, i % R))
        for i in range(5):
            one(w2, i % 2)
        for i in range(5):
            seen.append(one(w, (i + 1) % R))
        assert T.ERR_ASSERT in seen and T.ERR_BAD_SIGNATURE in seen and seen.count(T.OK) >= 8
    _lib.check(L.bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(1)))


def test_trim_releases_host_tier_state_and_the_next_call_rebuilds_it():
    """bsx_trim gives the scratch arena (grown by a witness request), the persistent key table and a captured graph back; the
    next requests — with and without graph replay — re-create them and still equal the oracle; misuse (null context) is an error."""
    import ctypes as C
    J, B, V = 4, 16, 10
    w = synth.Workload(79, 2, J, B, v=V)
    L, ctx = _lib.lib(), _lib.context(0)
    circ = CombinedSkipCircuit(V, J, B)

    def one(r, want_witness=False):
        S = int(w.first_height[r])
        _, want_out, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], S, int(w.latest[r]), w.validators[r], w.trusted[r], want_witness=want_witness)
        out, _, wit = circ.prove(w.input48(r), InputDataFetcher(w.headers[r], S, int(w.latest[r])), w.validators[r], w.trusted[r], want_witness=want_witness)
        assert out == want_out
        if want_witness:
            full = oracle.expand_range_witness(J, B, cw, v_max=V)
            assert np.asarray(wit).shape == full.shape and (np.asarray(wit) == full).all()

    one(0, want_witness=True)
    freed = C.c_uint64(0)
    _lib.check(L.bsx_trim(ctx, C.byref(freed)))
    assert freed.value >= J * int(T.map_layout(B)["n_elements"]) * 8       # at least the witness the arena held
    _lib.check(L.bsx_set_tuning(ctx, C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(1)))
    for i in range(6):                                                      # capture + replays on the re-created state
        one(i % 2)
    _lib.check(L.bsx_trim(ctx, C.byref(freed)))                             # drops the captured graph too
    assert freed.value > 0
    _lib.check(L.bsx_set_tuning(ctx, C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(0)))
    one(1)
    _lib.check(L.bsx_trim(ctx, None))
    assert L.bsx_trim(None, None) == T.ERR_BAD_ARG
