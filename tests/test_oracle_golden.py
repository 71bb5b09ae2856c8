"""CPU suite: the oracle (oracle/, plain C) against every golden vector the reference's own tests hold for the
header_range path, i.e. tests/golden/mocha4.json derived from circuits/fixtures/mocha-4.

Mirrors the reference's tests: test_get_data_commitment (circuits/builder.rs:488-529), test_prove_header_chain
(builder.rs:531-564), test_encode_data_root_tuple (builder.rs:566-608), and the fixture facts of SURVEY Appendix A.
"""
import numpy as np

import oracle
from blobstreamx_amd import types as T


def test_header_hashes_and_proofs(golden, mocha):
    hs, dh, lb = oracle.header_hashes(mocha["headers"])
    for i, h in enumerate(mocha["heights"]):
        b = golden["blocks"][str(h)]
        assert hs[i].tobytes().hex() == b["header_hash"]
        assert [bytes(a).hex() for a in dh[i]["aunts"]] == b["data_hash_proof"]["aunts"]
        assert bytes(dh[i]["leaf"]).hex() == b["data_hash_proof"]["leaf"]
        assert [bytes(a).hex() for a in lb[i]["aunts"]] == b["last_block_id_proof"]["aunts"]
        assert bytes(lb[i]["leaf"]).hex() == b["last_block_id_proof"]["leaf"]


def test_chain_links(mocha):
    _, _, lb = oracle.header_hashes(mocha["headers"])
    for i in range(1, 5):
        assert bytes(lb[i]["leaf"][2:34]) == mocha["hashes"][i - 1]


def test_encode_data_root_tuple_kat(golden):
    # circuits/builder.rs:584-605
    k = golden["kats"]["encode_data_root_tuple"]
    assert oracle.encode_data_root_tuple(bytes.fromhex(k["data_hash"]), k["height"]).hex() == k["expected"]


def test_get_data_commitment_fixture_answers(golden, mocha):
    # circuits/builder.rs:488-529 with MAX_LEAVES = 4, all four node-reported commitments
    for name, want in golden["data_commitments"].items():
        s, e = map(int, name.split("-"))
        rc, inp = oracle.data_commitment_inputs(mocha["headers"], 10000, mocha["latest"], s, e, 4)
        assert rc == T.OK
        assert inp["expected_data_commitment"].hex() == want
        dhs = np.stack([p["leaf"][2:34] for p in inp["data_hash_proofs"]])
        rc, root, af = oracle.get_data_commitment(dhs, s, e)
        assert rc == T.OK and af == 0 and root.hex() == want


def test_prove_header_chain(golden, mocha):
    # circuits/builder.rs:531-564: MAX_LEAVES = 4, blocks 10000..10004 through prove_subchain
    rc, inp = oracle.data_commitment_inputs(mocha["headers"], 10000, mocha["latest"], 10000, 10004, 4)
    assert rc == T.OK
    assert inp["start_header"] == mocha["hashes"][0] and inp["end_header"] == mocha["hashes"][4]
    rc, rec, _ = oracle.prove_subchain(4, inp["start_header"], inp["end_header"], inp["data_hash_proofs"],
                                       inp["last_block_id_proofs"], 10000, 10004, 10004, inp["end_header"])
    assert rc == T.OK and rec["assert_fail"] == 0 and rec["is_enabled"] == 1
    assert bytes(rec["data_merkle_root"]).hex() == golden["data_commitments"]["10000-10004"]
    assert bytes(rec["end_header"]) == mocha["hashes"][4] and rec["end_block"] == 10004
    assert bytes(rec["start_header"]) == mocha["hashes"][0] and rec["start_block"] == 10000


def test_prove_data_commitment_shapes(golden, mocha):
    hh = mocha["hashes"]
    for (J, B, s, e) in [(2, 2, 10000, 10004), (4, 4, 10000, 10002), (2, 8, 10002, 10004), (1, 4, 10000, 10004),
                         (4, 1, 10000, 10004), (8, 2, 10000, 10001)]:
        ctx = oracle.make_ctx(s, hh[s - 10000], e, hh[e - 10000])
        rc, out = oracle.prove_data_commitment(J, B, ctx, mocha["headers"], 10000, mocha["latest"])
        assert rc == T.OK, (J, B, s, e, hex(out["status"]))
        assert out["data_commitment"].hex() == golden["data_commitments"][f"{s}-{e}"]


def test_prove_data_commitment_rejects(mocha):
    hh = mocha["hashes"]
    # wrong end header -> A5/A9
    ctx = oracle.make_ctx(10000, hh[0], 10004, hh[3])
    rc, out = oracle.prove_data_commitment(2, 2, ctx, mocha["headers"], 10000, mocha["latest"])
    assert rc == T.ERR_ASSERT and out["status"] & (T.A5_END_HEADER | T.A9_FINAL)
    # wrong start header -> A3 (first slot) and A9
    ctx = oracle.make_ctx(10000, hh[1], 10004, hh[4])
    rc, out = oracle.prove_data_commitment(2, 2, ctx, mocha["headers"], 10000, mocha["latest"])
    assert rc == T.ERR_ASSERT and out["status"] & T.A9_FINAL
    # range longer than J*B -> A7 (builder.rs:292-297)
    ctx = oracle.make_ctx(10000, hh[0], 10004, hh[4])
    rc, out = oracle.prove_data_commitment(1, 2, ctx, mocha["headers"], 10000, mocha["latest"])
    assert rc == T.ERR_ASSERT and out["status"] & T.A7_RANGE


def test_next_header(golden, mocha):
    # circuits/builder.rs:411-443; SURVEY §3.4 fixture answer 10000 -> 10001
    rc, dc = oracle.prove_next_header_data_commitment(10000, mocha["hashes"][0], 10001, mocha["headers"][0], mocha["latest"])
    assert rc == T.OK and dc.hex() == golden["data_commitments"]["10000-10001"]
    rc, _ = oracle.prove_next_header_data_commitment(10000, mocha["hashes"][1], 10001, mocha["headers"][0], mocha["latest"])
    assert rc == T.ERR_ASSERT


def test_commits(golden, mocha):
    for i, h in enumerate(mocha["heights"]):
        b = golden["blocks"][str(h)]
        res, ok = oracle.verify_commit(mocha["commits"][i], mocha["hashes"][i])
        assert bytes(res["validators_hash"]).hex() == b["validators_hash"]
        assert res["n_enabled"] == 2 and res["n_signed"] == 2
        assert res["n_bad_signature"] == 0 and res["n_bad_message"] == 0 and res["two_thirds_ok"] == 1
        assert res["total_power"] == 50_000_000 and res["signed_power"] == 50_000_000
        assert list(ok) == [1, 1, 0, 0]
        _, dig = oracle.sha512_challenge(mocha["commits"][i][:2])
        for s in b["commit"]["signatures"]:
            assert dig[s["validator_index"]].tobytes().hex() == s["sha512_rAM"]
        # wrong header hash -> message check fails, signature still valid
        res, ok = oracle.verify_commit(mocha["commits"][i], mocha["hashes"][(i + 1) % 5])
        assert res["n_bad_message"] == 2 and res["signed_power"] == 0 and res["two_thirds_ok"] == 0
        # flipped signature bit
        bad = mocha["commits"][i].copy()
        bad[0]["signature"][3] ^= 0x10
        res, ok = oracle.verify_commit(bad, mocha["hashes"][i])
        assert res["n_bad_signature"] == 1 and res["first_bad_signature"] == 0 and list(ok) == [0, 1, 0, 0]
        assert res["two_thirds_ok"] == 0   # 1/2 of the power is not > 2/3


def test_witness_expansion_matches_layout(mocha):
    hh = mocha["hashes"]
    ctx = oracle.make_ctx(10000, hh[0], 10004, hh[4])
    rc, out = oracle.prove_data_commitment(2, 2, ctx, mocha["headers"], 10000, mocha["latest"], want_witness=True)
    assert rc == T.OK
    ml, rl = T.map_layout(2), T.reduce_layout()
    w = oracle.expand_range_witness(2, 2, out["compact"])
    assert w.size == 2 * int(ml["n_elements"]) + int(rl["n_elements"])
    assert w.max() < 2 ** 32   # every element is a bit, a bool or a u32 limb: canonical Goldilocks
    # first 256 elements = ctx.start_header_hash bits MSB first
    bits = np.unpackbits(np.frombuffer(hh[0], np.uint8))
    assert (w[:256] == bits).all()
    # the reduce node's last 32 bytes are the range commitment
    red = w[2 * int(ml["n_elements"]):]
    root_bits = red[96 * 8:128 * 8].astype(np.uint8)
    assert np.packbits(root_bits).tobytes() == out["data_commitment"]
    # words: ctx.start_block limbs lo, hi
    words = w[8 * int(ml["n_bytes"]):8 * int(ml["n_bytes"]) + int(ml["n_words"])]
    assert words[0] == 10000 and words[1] == 0 and words[2] == 10004


def test_next_header_reproduces_the_fixture_chain(golden, mocha):
    """CombinedStepCircuit::define (circuits/next_header.rs:25-46) over the mocha-4 fixture blocks: every step
    10000 -> 10001 ... 10003 -> 10004 verifies (two real signatures per commit) and yields next header hash ‖ the
    reference's own data commitment for [h, h+1) where the fixtures hold one (tests/golden/mocha4.json)."""
    for k in range(4):
        h = 10000 + k
        inp = h.to_bytes(8, "big") + mocha["hashes"][k]
        rc, out, cr = oracle.next_header(inp, mocha["headers"][k], mocha["headers"][k + 1], mocha["latest"], mocha["commits"][k + 1], chain_id=b"mocha-4")
        assert rc == 0, (h, rc)
        assert out[:32] == mocha["hashes"][k + 1]
        want = golden["data_commitments"].get(f"{h}-{h + 1}")
        if want:
            assert out[32:].hex() == want
        assert cr["n_signed"] == 2 and cr["two_thirds_ok"] == 1
    # wrong public input, a commit for the wrong block, a broken signature
    bad = (10000).to_bytes(8, "big") + mocha["hashes"][1]
    assert oracle.next_header(bad, mocha["headers"][0], mocha["headers"][1], mocha["latest"], mocha["commits"][1], chain_id=b"mocha-4")[0] == T.ERR_ASSERT
    inp = (10000).to_bytes(8, "big") + mocha["hashes"][0]
    assert oracle.next_header(inp, mocha["headers"][0], mocha["headers"][1], mocha["latest"], mocha["commits"][2], chain_id=b"mocha-4")[0] == T.ERR_BAD_SIGNATURE
    assert oracle.next_header(inp, mocha["headers"][0], mocha["headers"][2], mocha["latest"], mocha["commits"][2], chain_id=b"mocha-4")[0] == T.ERR_ASSERT
    v = mocha["commits"][1].copy()
    v[0]["signature"][5] ^= 1
    assert oracle.next_header(inp, mocha["headers"][0], mocha["headers"][1], mocha["latest"], v, chain_id=b"mocha-4")[0] == T.ERR_BAD_SIGNATURE
