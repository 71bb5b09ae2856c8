"""Wire-format ingest.  CPU part: the reference's fixture files, VERBATIM
(tests/golden/mocha-4/** are copies of circuits/fixtures/mocha-4 data files), decode to exactly the golden encoded
fields / sign-bytes.  GPU part: fixture JSON -> HIP path end to end, mirroring the reference's fixture-mode tests
test_get_data_commitment / test_prove_header_chain (circuits/builder.rs:488-564)."""
import os

import numpy as np
import pytest

from blobstreamx_amd import _lib
from blobstreamx_amd import ingest
from blobstreamx_amd import types as T

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mocha-4")


def test_signed_block_fixture_decodes_to_golden_fields(golden):
    f = ingest.FixtureFetcher(FIX)
    for h in range(10000, 10005):
        sb = f.signed_block(h)
        b = golden["blocks"][str(h)]
        assert sb["height"] == h and sb["n_validators"] == 2
        assert [x.hex() for x in T.header_fields(sb["header"])] == b["fields"]
        assert sb["block_hash"].hex() == b["header_hash"]
        for i, v in enumerate(b["validators"]):
            assert bytes(sb["validators"][i]["pubkey"]).hex() == v["pubkey"]
            assert sb["validators"][i]["voting_power"] == v["power"] and sb["validators"][i]["enabled"] == 1
        for s in b["commit"]["signatures"]:
            v = sb["validators"][s["validator_index"]]
            assert v["is_signed"] == 1
            assert bytes(v["signature"]).hex() == s["signature"]
            assert bytes(v["message"][:v["message_len"]]).hex() == s["sign_bytes"]
        assert sb["validators"][2]["enabled"] == 0 and sb["validators"][3]["enabled"] == 0
        hdr, height = ingest.header_from_json(open(os.path.join(FIX, str(h), "header.json")).read())
        assert height == h and hdr.tobytes() == sb["header"].tobytes()


def test_data_commitment_fixtures(golden):
    f = ingest.FixtureFetcher(FIX)
    for name, want in golden["data_commitments"].items():
        s, e = map(int, name.split("-"))
        assert f.get_data_commitment(s, e).hex() == want
    assert f.get_data_commitment(10002, 10002) == bytes(32)      # circuits/input.rs:70-72


def test_ingest_rejects_garbage():
    for bad in ["", "{", '{"result": {}}', '{"result": {"header": {"chain_id": 5}}}']:
        with pytest.raises(_lib.BsxError):
            ingest.header_from_json(bad)
    blk = open(os.path.join(FIX, "10000", "signed_block.json")).read()
    with pytest.raises(_lib.BsxError) as ei:
        ingest.signed_block_from_json(blk, 1)        # 2 validators do not fit MAX_VALIDATOR_SET_SIZE = 1
    assert ei.value.status == T.ERR_RANGE_TOO_LONG
    with pytest.raises(_lib.BsxError):
        ingest.signed_block_from_json(blk.replace("2023-09-07T12:45:59.767207173Z", "yesterday"), 4)


def test_ingest_commit_height_round_flag_and_int64_rules():
    """a missing / mismatching commit.height, a malformed round or block_id_flag and integers beyond int64
    are ingest errors (BSX_ERR_BAD_ARG) — not silently-zero sign-bytes that later look like bad signatures; a commit with
    round != 0 ingests to sign-bytes with the round field (block hash at offset 25)."""
    import json
    blk = json.load(open(os.path.join(FIX, "10000", "signed_block.json")))
    sb = blk["result"]["signed_header"] if "signed_header" in blk.get("result", {}) else blk.get("result", blk)
    commit = sb["commit"] if "commit" in sb else blk["result"]["commit"]

    def mutate(fn):
        b = json.loads(json.dumps(blk))
        r = b.get("result", b)
        c = (r["signed_header"] if "signed_header" in r else r)["commit"]
        fn(c, r)
        return json.dumps(b)
    ok = ingest.signed_block_from_json(json.dumps(blk), 4)
    assert ok["n_validators"] == 2
    for bad in (mutate(lambda c, r: c.pop("height")), mutate(lambda c, r: c.__setitem__("height", "10001")),
                mutate(lambda c, r: c.__setitem__("height", "-1")), mutate(lambda c, r: c.__setitem__("round", "x")),
                mutate(lambda c, r: c["signatures"][0].__setitem__("block_id_flag", 7)),
                mutate(lambda c, r: c["signatures"][0].pop("block_id_flag")),
                mutate(lambda c, r: c.__setitem__("height", "18446744073709551617"))):
        with pytest.raises(_lib.BsxError) as ei:
            ingest.signed_block_from_json(bad, 4)
        assert ei.value.status == T.ERR_BAD_ARG
    big = mutate(lambda c, r: r["validator_set"]["validators"][0].__setitem__("voting_power", "9223372036854775808"))
    with pytest.raises(_lib.BsxError):
        ingest.signed_block_from_json(big, 4)
    rnd = ingest.signed_block_from_json(mutate(lambda c, r: c.__setitem__("round", 5)), 4)
    v = rnd["validators"][rnd["validators"]["is_signed"] != 0][0]
    m = bytes(v["message"][:v["message_len"]])
    assert m[12] == 0x19 and m[13:21] == (5).to_bytes(8, "little") and m[25:57] == bytes(rnd["block_hash"])
    nil = ingest.signed_block_from_json(mutate(lambda c, r: c["signatures"][0].__setitem__("block_id_flag", 3)), 4)
    assert nil["validators"]["is_signed"].sum() == ok["validators"]["is_signed"].sum() - 1      # a NIL vote is not a signature


def test_time_and_varint_edge_cases():
    import json
    blk = json.load(open(os.path.join(FIX, "10000", "signed_block.json")))
    hdr = blk["result"]["header"]
    for t, secs, nanos in [("1970-01-01T00:00:00Z", 0, 0), ("2000-02-29T23:59:59.5Z", 951868799, 500000000),
                           ("2024-12-31T00:00:00.000000001Z", 1735603200, 1), ("2262-04-11T23:47:16.854775807Z", 9223372036, 854775807)]:
        hdr["time"] = t
        h, _ = ingest.header_from_json(json.dumps(blk))

        def varint(n):
            out = bytearray()
            while n >= 0x80:
                out.append((n & 0x7f) | 0x80)
                n >>= 7
            out.append(n)
            return bytes(out)
        want = (b"\x08" + varint(secs) if secs else b"") + (b"\x10" + varint(nanos) if nanos else b"")
        assert T.header_fields(h)[3] == want, t


@pytest.mark.gpu
def test_fixture_json_to_gpu_end_to_end(golden):
    """JSON files -> ingest -> HIP path: header hashes equal the node's own block ids, the commit verifies, the
    node-reported data commitments come out, next_header and header_range produce the fixture answers."""
    from blobstreamx_amd.builder import CombinedSkipCircuit, DataCommitmentBuilder, InputDataFetcher, verify_commits
    f = ingest.FixtureFetcher(FIX, v_max=4)
    blocks = [f.signed_block(h) for h in range(10000, 10005)]
    headers = np.array([b["header"] for b in blocks], dtype=T.HEADER)
    fetcher = InputDataFetcher(headers, 10000, 10006)
    hashes = fetcher.header_hashes()
    for i, b in enumerate(blocks):
        assert hashes[i].tobytes() == b["block_hash"]                      # our hash == commit.block_id.hash from the node
    res, ok = verify_commits(np.stack([b["validators"] for b in blocks]), hashes)
    assert (res["n_bad_signature"] == 0).all() and (res["n_bad_message"] == 0).all() and (res["two_thirds_ok"] == 1).all()
    for i, b in enumerate(blocks):
        assert bytes(res[i]["validators_hash"]) == bytes(b["header"]["hash"][2][2:34])
    bld = DataCommitmentBuilder()
    for (s, e) in [(10000, 10001), (10000, 10002), (10000, 10004), (10002, 10004)]:
        out = bld.prove_data_commitment(fetcher, 2, 2, s, hashes[s - 10000].tobytes(), e, hashes[e - 10000].tobytes())
        assert out["data_commitment"] == f.get_data_commitment(s, e)
    assert bld.prove_next_header_data_commitment(fetcher, 10000, hashes[0].tobytes(), 10001) == f.get_data_commitment(10000, 10001)
    # header_range 10000 -> 10004: trusted set = validators of 10000 (unchanged on mocha-4 here), target commit = block 10004's
    inp = (10000).to_bytes(8, "big") + hashes[0].tobytes() + (10004).to_bytes(8, "big")
    assert inp.hex() == golden["kats"]["header_range_input_10000_10004"]
    trusted = blocks[0]["validators"].copy()
    trusted["is_signed"] = 0
    out, cres, _ = CombinedSkipCircuit(4, 2, 2, chain_id=b"mocha-4").prove(inp, fetcher, blocks[4]["validators"], trusted)
    assert out[:32] == hashes[4].tobytes() and out[32:] == f.get_data_commitment(10000, 10004)
    assert cres["trusted_signed_power"] == 50_000_000


@pytest.mark.gpu
def test_cli_prove_input_json(tmp_path, golden):
    """`<circuit> prove input.json` (succinct.json:5-46): packed EVM input in, abi.encode(bytes32, bytes32) out."""
    import json
    from blobstreamx_amd import cli
    h0 = golden["blocks"]["10000"]["header_hash"]
    inp = tmp_path / "input.json"
    out = tmp_path / "output.json"
    wit = tmp_path / "w.bin"
    inp.write_text(json.dumps({"type": "req_bytes", "releaseId": "x", "data": {"input": "0x" + golden["kats"]["header_range_input_10000_10004"]}}))
    caps = tmp_path / "caps.json"
    assert cli.main(["header_range_mocha", "prove", str(inp), "--fixtures", FIX, "--jobs", "2", "--batch", "2", "--validators", "4",
                     "--output", str(out), "--witness", str(wit), "--caps", str(caps)]) == 0
    import oracle
    cj = json.load(open(caps))
    wv = np.fromfile(wit, dtype="<u8")
    nel = int(T.map_layout(2)["n_elements"])
    n_leaves = 1
    while n_leaves * 135 < nel:
        n_leaves *= 2
    _, cap = oracle.poseidon_merkle_tree(wv[nel:2 * nel], 135, n_leaves, min(4, n_leaves.bit_length() - 1))
    assert cj["map_jobs"][1] == ["0x" + "".join(f"{int(x):016x}" for x in d) for d in cap]
    got = json.load(open(out))["data"]["output"]
    assert got == "0x" + golden["blocks"]["10004"]["header_hash"] + golden["data_commitments"]["10000-10004"]
    # the whole circuit's witness: 2 map jobs, 1 reduce node, the COMMIT unit of block 10004's commit, the SKIP unit (4 slots)
    assert wit.stat().st_size == 8 * T.header_range_witness_elements(2, 2, 4)
    assert set(cj["cap_height"]) == {"map_jobs", "reduce_nodes", "commit", "skip"} and len(cj["commit"]) == 1 and len(cj["skip"]) == 1
    cl = T.commit_layout(4)
    off = 2 * nel + int(T.reduce_layout()["n_elements"])
    unit = wv[off:off + int(cl["n_elements"])]
    nl = 1
    while nl * 135 < unit.size:
        nl *= 2
    _, cap = oracle.poseidon_merkle_tree(unit, 135, nl, cj["cap_height"]["commit"])
    assert cj["commit"][0] == ["0x" + "".join(f"{int(x):016x}" for x in d) for d in cap]
    inp.write_text(json.dumps({"type": "req_bytes", "data": {"input": "0x" + (10000).to_bytes(8, "big").hex() + h0}}))
    assert cli.main(["next_header_mocha", "prove", str(inp), "--fixtures", FIX, "--validators", "4", "--output", str(out), "--witness", str(wit)]) == 0
    assert wit.stat().st_size == 8 * T.next_header_witness_elements(4)
    got = json.load(open(out))["data"]["output"]
    assert got == "0x" + golden["blocks"]["10001"]["header_hash"] + golden["data_commitments"]["10000-10001"]
