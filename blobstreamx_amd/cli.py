"""`<circuit> build | prove input.json` — the process boundary of the reference's entrypoint binaries
(bin/header_range_{1024,2048,mocha}.rs:6-17, bin/next_header{,_mocha}.rs:5-8 via plonky2x `Plonky2xFunction::entrypoint`;
`succinct.json:5-46` runs `./build/<circuit> prove input.json`), SURVEY §8f rank 2.

    python -m blobstreamx_amd.cli header_range_2048 prove input.json --fixtures DIR [--output output.json] [--witness w.bin]
    python -m blobstreamx_amd.cli next_header prove input.json --fixtures DIR
    python -m blobstreamx_amd.cli header_range_mocha build

input.json  [UPSTREAM plonky2x ProofRequest, bytes flavour]: {"type": "req_bytes", "data": {"input": "0x<hex>"}} with
            48 bytes abi.encodePacked(uint64 trusted_block, bytes32 trusted_header_hash, uint64 target_block) for
            header_range (circuits/header_range.rs:33-35, contracts/src/BlobstreamX.sol:142-146) or 40 bytes
            (uint64 prev_block, bytes32 prev_header_hash) for next_header (circuits/next_header.rs:26-27).
output.json {"type": "res_bytes", "data": {"output": "0x<64 bytes>"}} = abi.encode(bytes32, bytes32)
            (header_range.rs:57-58, BlobstreamX.sol:155-158).  No proof is produced: this engine generates the WITNESS
            (optionally dumped as little-endian u64 Goldilocks elements); proving stays with plonky2x.
Chain data comes from a fixture directory in the reference's own layout (`InputDataMode::Fixture`,
circuits/input.rs:97-101): <dir>/<height>/signed_block.json.
"""
import argparse
import json
import os
import sys

import numpy as np

from . import ingest
from . import types as T

# const-generic instantiations of the reference's bins: (MAX_VALIDATOR_SET_SIZE, NB_MAP_JOBS, BATCH_SIZE)
CIRCUITS = {
    "header_range_1024": (100, 32, 32),    # bin/header_range_1024.rs:7-9
    "header_range_2048": (100, 32, 64),    # bin/header_range_2048.rs:7-9
    "header_range_mocha": (100, 32, 32),   # bin/header_range_mocha.rs (Mocha4BlobstreamXConfig1024, config.rs:22-28)
    "next_header": (100, 1, 1),            # bin/next_header.rs:5-8
    "next_header_mocha": (100, 1, 1),
}


def _read_input(path):
    req = json.load(open(path))
    data = req.get("data", req)
    hexs = data["input"]
    return bytes.fromhex(hexs[2:] if hexs.startswith("0x") else hexs)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="blobstreamx_amd.cli")
    ap.add_argument("circuit", choices=sorted(CIRCUITS))
    ap.add_argument("command", choices=["build", "prove"])
    ap.add_argument("input", nargs="?")
    ap.add_argument("--fixtures", help="fixture directory (<dir>/<height>/signed_block.json)")
    ap.add_argument("--latest", type=int, help="chain head (default: highest fixture height + 2)")
    ap.add_argument("--jobs", type=int)
    ap.add_argument("--batch", type=int)
    ap.add_argument("--validators", type=int)
    ap.add_argument("--output", default="output.json")
    ap.add_argument("--witness")
    ap.add_argument("--caps", help="also write the Poseidon Merkle caps of the witness columns (one per map job / reduce node; "
                                   "PoseidonGoldilocksConfig, rows of 135 elements, cap height 4) to this JSON file")
    ap.add_argument("--chain-id", help="C::CHAIN_ID_BYTES (default: mocha-4 for the *_mocha circuits, celestia otherwise; config.rs:6-28)")
    a = ap.parse_args(argv)
    chain_id = (a.chain_id or ("mocha-4" if a.circuit.endswith("_mocha") else "celestia")).encode()
    V, J, B = CIRCUITS[a.circuit]
    V, J, B = a.validators or V, a.jobs or J, a.batch or B
    if a.command == "build":
        # the reference compiles circuits here; the witness engine has nothing to build beyond its layout
        nel = T.next_header_witness_elements(V) if a.circuit.startswith("next_header") else T.header_range_witness_elements(J, B, V)
        print(json.dumps({"circuit": a.circuit, "max_validators": V, "nb_map_jobs": J, "batch_size": B, "witness_elements": int(nel)}))
        return 0
    if not a.input or not a.fixtures:
        ap.error("prove needs input.json and --fixtures")
    from .builder import CombinedStepCircuit, CombinedSkipCircuit, DataCommitmentBuilder, InputDataFetcher, verify_commits
    inp = _read_input(a.input)
    fx = ingest.FixtureFetcher(a.fixtures, v_max=V)
    if a.circuit.startswith("header_range"):
        if len(inp) != 48:
            raise SystemExit("header_range input must be 48 bytes (uint64 ‖ bytes32 ‖ uint64)")
        trusted, target = int.from_bytes(inp[:8], "big"), int.from_bytes(inp[40:], "big")
        # The hint serves every batch up to min(batch_end, latest - 2) (circuits/input.rs:160-165): headers past the target are
        # loaded while the fixture directory has them, so that the disabled slots of the witness carry the same real headers
        # the reference's hint would fetch; where the fixtures end, `latest` is clamped accordingly (those slots are then zero
        # padded — the public output does not depend on it, the disabled-slot witness content does).
        blocks = {h: fx.signed_block(h) for h in range(trusted, target + 1)}
        last = target
        want_last = min(trusted + J * B, (a.latest - 2) if a.latest else trusted + J * B)
        while last < want_last and os.path.exists(os.path.join(a.fixtures, str(last + 1), "signed_block.json")):
            last += 1
            blocks[last] = fx.signed_block(last)
        latest = min(a.latest, last + 2) if a.latest else last + 2
        if a.latest and latest != a.latest:
            print(f"note: fixtures end at {last}; chain head clamped from {a.latest} to {latest}", file=sys.stderr)
        headers = np.array([blocks[h]["header"] for h in range(trusted, last + 1)], dtype=T.HEADER)
        fetcher = InputDataFetcher(headers, trusted, latest)
        tr = blocks[trusted]["validators"].copy()
        tr["is_signed"] = 0
        out, commit, wit = CombinedSkipCircuit(V, J, B, chain_id=chain_id).prove(inp, fetcher, blocks[target]["validators"], tr,
                                                              want_witness=bool(a.witness or a.caps))
    else:
        if len(inp) != 40:
            raise SystemExit("next_header input must be 40 bytes (uint64 ‖ bytes32)")
        prev = int.from_bytes(inp[:8], "big")
        blocks = {h: fx.signed_block(h) for h in (prev, prev + 1)}
        headers = np.array([blocks[prev]["header"], blocks[prev + 1]["header"]], dtype=T.HEADER)
        # CombinedStepCircuit::define (circuits/next_header.rs:25-46) behind the C ABI (bsx_next_header)
        out, _, wit = CombinedStepCircuit(blocks[prev + 1]["validators"].size, chain_id=chain_id).prove(
            inp, headers[0], headers[1], a.latest or prev + 3, blocks[prev + 1]["validators"], want_witness=True)
        V = blocks[prev + 1]["validators"].size
    json.dump({"type": "res_bytes", "data": {"output": "0x" + out.hex()}}, open(a.output, "w"))
    if a.witness and wit is not None:
        wit.astype("<u8").tofile(a.witness)
    if a.caps and wit is not None:
        # what the prover commits to first: Merkle caps of the witness columns, one per unit of the witness
        from .poseidon import witness_leaf_count, witness_merkle_caps
        hexd = lambda caps_: [["0x" + "".join(f"{int(x):016x}" for x in d) for d in c] for c in caps_]
        ch = lambda lay: min(4, witness_leaf_count(int(lay["n_elements"]), 135).bit_length() - 1)
        units = ([("map_jobs", T.map_layout(B), J), ("reduce_nodes", T.reduce_layout(), J - 1), ("commit", T.commit_layout(V), 1),
                  ("skip", T.skip_layout(V), 1)] if a.circuit.startswith("header_range") else [("commit", T.commit_layout(V), 1), ("step", T.step_layout(), 1)])
        caps, off = {"leaf_len": 135, "cap_height": {}}, 0
        for uname, lay, n in units:
            nel = n * int(lay["n_elements"])
            caps["cap_height"][uname] = ch(lay)
            caps[uname] = hexd(witness_merkle_caps(lay, wit[off:off + nel], n, 135, ch(lay))) if n else []
            off += nel
        json.dump(caps, open(a.caps, "w"))
    print("0x" + out.hex())
    return 0


if __name__ == "__main__":
    sys.exit(main())
