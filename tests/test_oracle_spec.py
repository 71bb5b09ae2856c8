"""CPU suite: spec-level known answers + differential tests that pin the oracle's primitives where the
reference's fixtures cannot (SURVEY.md §8c "parity unpinned" list): FIPS 180-4 via hashlib, RFC 8032 §7.1,
and cross-implementation agreement oracle (5x51-bit C) / synth (16x16-bit C signer)."""
import hashlib
import random

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import types as T

RFC8032 = [
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
     "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
     "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
     "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
]
L = 2 ** 252 + 27742317777372353535851937790883648493


@pytest.mark.parametrize("portable", [False, True])
def test_sha_differential(portable):
    rnd = random.Random(7)
    oracle.sha256_force_portable(portable)
    try:
        for n in list(range(0, 260)) + [511, 512, 1000, 4097]:
            m = bytes(rnd.getrandbits(8) for _ in range(n))
            assert oracle.sha256(m) == hashlib.sha256(m).digest()
            assert oracle.sha512(m) == hashlib.sha512(m).digest()
    finally:
        oracle.sha256_force_portable(False)


def test_synth_hashes():
    rnd = random.Random(9)
    for n in list(range(0, 140)) + [255, 256, 300]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        assert synth.sha256(m) == hashlib.sha256(m).digest()
        assert synth.sha512(m) == hashlib.sha512(m).digest()


def test_rfc8032_vectors():
    for sk, pk, m, sig in RFC8032:
        sk, pk, m, sig = map(bytes.fromhex, (sk, pk, m, sig))
        assert oracle.ed25519_verify(pk, m, sig)
        assert synth.ed25519_keypair(sk) == pk
        assert synth.ed25519_sign(sk, m) == sig
        bad = bytearray(sig)
        bad[5] ^= 1
        assert not oracle.ed25519_verify(pk, m, bytes(bad))
        assert not oracle.ed25519_verify(pk, m + b"x", sig)
        # s + L is the same scalar mod L but non-canonical: must be rejected (s < L rule)
        s = int.from_bytes(sig[32:], "little") + L
        if s < 2 ** 256:
            assert not oracle.ed25519_verify(pk, m, sig[:32] + s.to_bytes(32, "little"))


def test_ed25519_rejects_bad_encodings():
    sk, pk, m, sig = map(bytes.fromhex, RFC8032[2])
    # non-canonical y (y >= p) in A and in R
    assert not oracle.ed25519_verify((2 ** 255 - 1).to_bytes(32, "little"), m, sig)
    assert not oracle.ed25519_verify(pk, m, (2 ** 255 - 19 + 1).to_bytes(32, "little") + sig[32:])
    # y = 2 is not on the curve
    assert not oracle.ed25519_verify((2).to_bytes(32, "little"), m, sig)
    # x = 0 with sign bit set (y = 1 | 1<<255)
    assert not oracle.ed25519_verify((1 | 1 << 255).to_bytes(32, "little"), m, sig)


def test_sc_reduce_differential():
    rnd = random.Random(11)
    for x in [0, 1, L - 1, L, L + 1, 2 ** 512 - 1, 2 ** 252, 2 ** 256 - 1] + [rnd.getrandbits(512) for _ in range(200)]:
        assert int.from_bytes(oracle.sc_reduce64(x.to_bytes(64, "little")), "little") == x % L


def test_synth_signatures_verify_under_oracle():
    rnd = random.Random(13)
    for _ in range(20):
        seed = bytes(rnd.getrandbits(8) for _ in range(32))
        msg = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 125)))
        pk, sig = synth.ed25519_keypair(seed), synth.ed25519_sign(seed, msg)
        assert oracle.ed25519_verify(pk, msg, sig)
        if msg:
            assert not oracle.ed25519_verify(pk, msg[:-1] + bytes([msg[-1] ^ 1]), sig)


def test_merkle_against_hashlib():
    def root(items):
        if not items:
            return hashlib.sha256(b"").digest()
        if len(items) == 1:
            return hashlib.sha256(b"\x00" + items[0]).digest()
        k = 1
        while k * 2 < len(items):
            k *= 2
        return hashlib.sha256(b"\x01" + root(items[:k]) + root(items[k:])).digest()

    rnd = random.Random(17)
    # masked power-of-two tree (compute_root_from_leaves [UPSTREAM]) == RFC 6962 root over the enabled prefix
    for B in (1, 2, 4, 8, 16, 64):
        dhs = np.frombuffer(bytes(rnd.getrandbits(8) for _ in range(32 * B)), np.uint8).reshape(B, 32)
        for n in sorted({0, 1, B // 2, B - 1, B} & set(range(0, B + 1))):
            rc, got, af = oracle.get_data_commitment(dhs, 500, 500 + n)
            assert rc == T.OK
            tuples = [b"\x00" * 24 + (500 + i).to_bytes(8, "big") + dhs[i].tobytes() for i in range(n)]
            want = root(tuples) if n else hashlib.sha256(b"\x00" + b"\x00" * 24 + (500).to_bytes(8, "big") + dhs[0].tobytes()).digest()
            assert got == want, (B, n)
    rc, _, af = oracle.get_data_commitment(dhs, 10, 9)
    assert rc == T.ERR_ASSERT and af & T.A1_END_GTE_START


def test_synthetic_chain_is_consistent():
    w = synth.Workload(99, 2, 4, 8, v=5, v_max=8)
    hs, dh, lb = oracle.header_hashes(w.headers[1])
    assert (hs == w.hashes[1]).all()
    for i in range(1, w.hpr):
        assert bytes(lb[i]["leaf"][2:34]) == hs[i - 1].tobytes()
    res, ok = oracle.verify_commit(w.validators[1], w.hashes[1, w.n_blocks])
    assert res["n_enabled"] == 5 and res["n_signed"] == 5 and res["n_bad_signature"] == 0 and res["n_bad_message"] == 0
    assert bytes(res["validators_hash"]) == w.valset.hash.tobytes()
    assert w.headers[1][3]["hash"][2][2:34].tobytes() == w.valset.hash.tobytes()
    rc, out, cres, _ = oracle.header_range(4, 8, w.input48(1), w.headers[1], int(w.first_height[1]), int(w.latest[1]),
                                           w.validators[1], w.trusted[1])
    assert rc == T.OK and out[:32] == w.hashes[1, 32].tobytes()
    assert cres["two_thirds_ok"] == 1 and cres["trusted_signed_power"] == cres["total_power"]


def test_chain_id_is_checked_by_skip_and_step():
    """builder.skip / builder.step take C::CHAIN_ID_BYTES (header_range.rs:42-43, next_header.rs:32-33): a fork that
    carries another chain id — same heights, same validator keys, validly signed — must be rejected, also when the
    other id merely has the circuit's id as a prefix or differs in its last byte."""
    for fork_id in ("celestiaX", "celestib", "mocha-4", "c"):
        w = synth.Workload(98, 1, 2, 4, v=4, chain_id=fork_id)
        args = (2, 4, w.input48(0), w.headers[0], int(w.first_height[0]), int(w.latest[0]), w.validators[0], w.trusted[0])
        assert oracle.header_range(*args, chain_id=fork_id.encode())[0] == T.OK
        assert oracle.header_range(*args)[0] == T.ERR_ASSERT                      # circuit constant b"celestia"
        wS = synth.Workload(98, 1, 2, 4, v=4, chain_id=fork_id, mode="S")
        inp = int(wS.first_height[0]).to_bytes(8, "big") + wS.hashes[0, 0].tobytes()
        step = (inp, wS.headers[0, 0], wS.headers[0, 1], int(wS.latest[0]), wS.validators[0].reshape(-1, 4)[0])
        assert oracle.next_header(*step, chain_id=fork_id.encode())[0] == T.OK
        assert oracle.next_header(*step)[0] == T.ERR_ASSERT


def test_commit_round_nonzero_nil_and_absent_votes():
    """with a round field (0x19 + 8 bytes) the signed block hash sits at offset 25, not 16.  Validators that
    voted NIL (BlockIDFlagNil: a VALID signature over a vote without a block id) or are ABSENT have is_signed = 0 and must
    count for nothing, whatever bytes their slot carries.  Checked against an independent Python reading of the bytes."""
    w = synth.Workload(33, 2, 2, 4, v=16, round=3, nil_permille=250, absent_permille=150)
    for r in range(2):
        vals, hh = w.validators[r], w.commit_hashes[r].tobytes()
        res, ok = oracle.verify_commit(vals, hh)
        signed = vals["is_signed"] != 0
        assert 0 < signed.sum() < 16 and (vals["message_len"][~signed] < 60).all()          # nil votes are short, absent empty
        for v in vals[signed]:
            m = bytes(v["message"][:v["message_len"]])
            assert m[12] == 0x19 and m[13:21] == (3).to_bytes(8, "little") and m[25:57] == hh and m[16:48] != hh
            assert oracle.ed25519_verify(bytes(v["pubkey"]), m, bytes(v["signature"]))
        nil = (~signed) & (vals["message_len"] > 0)
        assert nil.any()
        for v in vals[nil]:                                                                # nil vote: valid signature, no block id
            m = bytes(v["message"][:v["message_len"]])
            assert oracle.ed25519_verify(bytes(v["pubkey"]), m, bytes(v["signature"])) and hh not in m
        assert (ok == signed).all() and res["n_signed"] == signed.sum() and res["n_bad_message"] == 0 and res["n_bad_signature"] == 0
        assert res["signed_power"] == vals["voting_power"][signed].sum() and res["total_power"] == vals["voting_power"].sum()
        # the hash check really looks at offset 25: a commit whose votes carry the hash at 16 (round 0) fails it when relabelled
        bad = vals.copy()
        k = int(np.nonzero(signed)[0][0])
        bad[k]["message"][25] ^= 1
        res2, ok2 = oracle.verify_commit(bad, hh)
        assert res2["n_bad_message"] == 1 and res2["n_bad_signature"] == 1 and ok2[k] == 0
    # end to end: header_range over a commit with round != 0 (2/3 may or may not hold: the verdict is the tally's)
    w2 = synth.Workload(34, 1, 2, 4, v=10, round=7)
    rc, out, cres, _ = oracle.header_range(2, 4, w2.input48(0), w2.headers[0], int(w2.first_height[0]), int(w2.latest[0]), w2.validators[0], w2.trusted[0])
    assert rc == T.OK and cres["n_signed"] == 10 and out[:32] == w2.hashes[0, 8].tobytes()


def test_voting_power_beyond_max_total_is_rejected():
    """powers that add up beyond Tendermint's MaxTotalVotingPower (MaxInt64 / 8) would wrap the u64 tallies and
    could pass the 2/3 and 1/3 rules spuriously: flagged (power_overflow) and turned into BSX_ERR_BAD_ARG."""
    w = synth.Workload(36, 1, 2, 4, v=6)
    vals = w.validators[0].copy()
    res, _ = oracle.verify_commit(vals, w.commit_hashes[0].tobytes())
    assert res["power_overflow"] == 0 and res["two_thirds_ok"] == 1
    big = vals.copy()
    big["voting_power"][:4] = (1 << 62)                    # 4 * 2^62 = 2^64: the u64 sum wraps to the two small powers
    big["is_signed"][:4] = 0                               # nobody big signs: without the guard 2/2 of the WRAPPED total signed
    res, _ = oracle.verify_commit(big, w.commit_hashes[0].tobytes())
    assert res["power_overflow"] == 1 and res["two_thirds_ok"] == 0 and res["total_power"] < (1 << 40)
    edge = vals.copy()
    edge["voting_power"] = 0
    edge["voting_power"][0] = 1152921504606846975          # exactly MaxTotalVotingPower is still fine
    assert oracle.verify_commit(edge, w.commit_hashes[0].tobytes())[0]["power_overflow"] == 0
    edge["voting_power"][1] = 1
    assert oracle.verify_commit(edge, w.commit_hashes[0].tobytes())[0]["power_overflow"] == 1
    args = (2, 4, w.input48(0), w.headers[0], int(w.first_height[0]), int(w.latest[0]))
    assert oracle.header_range(*args, big, w.trusted[0])[0] == T.ERR_BAD_ARG
    tr = w.trusted[0].copy()
    tr["voting_power"][:] = (1 << 61)
    assert oracle.header_range(*args, w.validators[0], tr)[0] == T.ERR_BAD_ARG
    start, commit = _skip_search_case()
    s2 = start.copy()
    s2["voting_power"][:] = (1 << 61)
    rc, _, ev = oracle.find_block_to_request(1000, 1002, s2, [1002], commit([0, 1, 2])[None])
    assert rc == T.ERR_BAD_ARG and ev["power_overflow"][0] == 1 and ev["valid"][0] == 0


def _skip_search_case():
    """Start set of 9 validators (total power 100) and commits with chosen signers."""
    V = 9
    start = synth.ValidatorSet(5, V).as_validators(V)
    start["voting_power"] = [10, 10, 10, 10, 10, 10, 10, 10, 20]

    def commit(signers, strangers=()):
        c = start.copy()
        c["is_signed"] = 0
        for i in signers:
            c[i]["is_signed"] = 1
        for i in strangers:                       # slot taken over by a key that is not in the start set, and signs
            c[i]["pubkey"] = np.frombuffer(bytes([i + 1]) * 32, np.uint8)
            c[i]["is_signed"] = 1
        return c
    return start, commit


def test_find_block_to_request_follows_the_reference_loop():
    """fetcher.rs:60-87: first valid candidate of the halving sequence, distance 1 accepted unconditionally.  The
    predicate is [UPSTREAM] (parity unpinned): > 1/3 of the start set's power among the validators that signed."""
    start, commit = _skip_search_case()
    S, M = 1000, 1100
    heights, c = [], M
    while c - S > 1:
        heights.append(c)
        c = (c + S) // 2
    assert heights == [1100, 1050, 1025, 1012, 1006, 1003]
    # overlap 30 (no), 30 with 50 of stranger power signing (no), 40 (yes), 100 (yes, never reached)
    cands = [commit([0, 1, 2]), commit([0, 1, 2], strangers=[3, 4, 5, 6, 7]), commit([0, 1, 8]), commit(range(9)),
             commit(range(9)), commit(range(9))]
    rc, blk, ev = oracle.find_block_to_request(S, M, start, heights, np.stack(cands))
    assert rc == 0 and blk == 1025
    assert list(ev["overlap_power"][:4]) == [30, 30, 40, 100] and list(ev["valid"][:4]) == [0, 0, 1, 1]
    assert ev["signed_power"][1] == 80 and ev["start_total_power"][0] == 100 and ev["target_total_power"][0] == 100
    # nothing valid: the loop ends at start + 1 without evaluating it
    none = np.stack([commit([0])] * len(heights))
    rc, blk, _ = oracle.find_block_to_request(S, M, start, heights, none)
    assert rc == 0 and blk == S + 1
    # a visited height missing from the candidates is an argument error
    rc, _, _ = oracle.find_block_to_request(S, M, start, heights[:1], none[:1])
    assert rc == T.ERR_BAD_ARG
    # exactly one third is NOT enough (strict inequality)
    start3 = start.copy()
    start3["voting_power"] = 10
    third = start3.copy()
    third["is_signed"] = 0
    third["is_signed"][:3] = 1
    rc, blk, ev = oracle.find_block_to_request(S, S + 2, start3, [S + 2], third[None])
    assert blk == S + 1 and ev["valid"][0] == 0 and ev["overlap_power"][0] * 3 == ev["start_total_power"][0]


def test_ed25519_public_vectors_parity_unpinned_by_reference():
    """tests/golden/ed25519_vectors.json (RFC 8032 §7.1 TEST 1-3 + SHA(abc); constructed edge cases with verdicts from the
    stdlib big-integer verifier of tests/golden/gen_ed25519_vectors.py): the oracle must give every listed verdict, and its
    SHA-512 must give the listed challenge digests."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ed25519_vectors.json")))
    assert len(d["rfc8032"]) == 4 and len(d["edge"]) >= 20
    for e in d["rfc8032"] + d["edge"]:
        pk, m, sig = (bytes.fromhex(e[k]) for k in ("public_key", "message", "signature"))
        assert oracle.ed25519_verify(pk, m, sig) == e["valid"], e["name"]
        if "sha512_challenge" in e:
            assert oracle.sha512(sig[:32] + pk + m).hex() == e["sha512_challenge"]
    assert oracle.sha512(b"abc").hex() == d["rfc8032"][3]["message"]
    assert any(e.get("discriminates") == "cofactorless" and not e["valid"] for e in d["edge"])
