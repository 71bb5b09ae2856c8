#!/usr/bin/env python3
"""Randomised soak of the commit check and the host tier against the oracle (a tool beside the suite, `python
tools/soak_commits.py 300`).  Round kinds: (S) mode S through bsx_dev_verify_commits — random commits x validators, nil /
absent votes, random corruptions (signature, public key, voting power, the signed flag, the header hash), sharded over a random
world: ok bits, every commit result, every rank's fold and the range verdict; (H) one proof request through bsx_header_range
(host pointers) with random shape and tampering, graph replay on or off: status code, 64-byte output, commit result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import oracle, synth
from blobstreamx_amd import _lib, types as T
from blobstreamx_amd.builder import CombinedSkipCircuit, InputDataFetcher
from blobstreamx_amd.stress import CommitShard, range_verdict


def clean(r):
    r = np.array(r).copy(); r["_pad"] = 0
    return r.tobytes()


def mode_s(rng):
    J = int(rng.choice([1, 2, 4, 8])); B = int(rng.choice([2, 8, 16, 32])); nh = J * B
    V = int(rng.choice([1, 2, 3, 7, 12, 33, 64, 100, 128])); world = int(rng.choice([w for w in (1, 2, 4, 8) if nh % w == 0]))
    w = synth.Workload(int(rng.integers(1, 1 << 20)), 1, J, B, v=V, mode="S", nil_permille=int(rng.choice([0, 0, 100, 300])),
                       absent_permille=int(rng.choice([0, 0, 50, 300])))
    vals = w.validators.reshape(nh, V).copy()
    hashes = w.commit_hashes.copy()
    what = []
    for _ in range(int(rng.integers(0, 6))):
        c, k, kind = int(rng.integers(0, nh)), int(rng.integers(0, V)), int(rng.integers(0, 5))
        if kind == 0: vals[c, k]["signature"][int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: vals[c, k]["pubkey"][int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2: vals[c, k]["voting_power"] = int(rng.integers(0, 1 << 40))
        elif kind == 3: vals[c, k]["is_signed"] ^= 1
        else: hashes[c, int(rng.integers(0, 32))] ^= 1
        what.append((c, k, kind))
    ref = [oracle.verify_commit(vals[c], hashes[c].tobytes()) for c in range(nh)]
    folds = []
    for g in range(world):
        sh = CommitShard(nh, V, rank=g, world=world)
        sh.upload(vals, hashes)
        for _ in range(int(rng.integers(1, 3))): sh.step()
        ok, res, fold = sh.download()
        for c in range(sh.n):
            rres, rok = ref[sh.first + c]
            assert (ok[c] == rok).all(), ("S ok bits", J, B, V, world, g, c, what)
            assert clean(res[c]) == clean(rres), ("S result", J, B, V, world, g, c, what)
        want = oracle.commit_fold(np.array([ref[sh.first + c][0] for c in range(sh.n)], T.COMMIT_RESULT), sh.first)
        assert fold.tobytes() == want.tobytes(), ("S fold", J, B, V, world, g, what)
        folds.append(fold)
        del sh
    v = range_verdict(np.array(folds, T.COMMIT_FOLD))
    good = [bool(r["two_thirds_ok"]) and not r["n_bad_signature"] and not r["n_bad_message"] and not r["power_overflow"] for r, _ in ref]
    assert v["commits"] == nh and v["ok"] == sum(good) and v["first_failing"] == (good.index(False) if False in good else None), ("S verdict", v, what)
    print(f"  S: {nh} commits x {V} validators, world {world}, corruptions {what} ok", flush=True)
    return nh


def host_tier(rng):
    J = int(rng.choice([1, 2, 4, 8, 16])); B = int(rng.choice([1, 2, 8, 16, 32]))
    if J * B < 2: return 0
    V = int(rng.choice([1, 2, 3, 10, 20, 100]))
    n_blocks = int(rng.integers(1, J * B + 1)) if rng.integers(0, 3) else J * B
    w = synth.Workload(int(rng.integers(1, 1 << 20)), 1, J, B, v=V, n_blocks=n_blocks, absent_permille=int(rng.choice([0, 0, 100])),
                       nil_permille=int(rng.choice([0, 0, 50])))
    what = []
    for _ in range(int(rng.integers(0, 3))):
        kind = int(rng.integers(0, 5))
        if kind == 0: w.headers[0, int(rng.integers(0, w.hpr))]["hash"][1][int(rng.integers(0, 30))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: w.headers[0, int(rng.integers(0, w.hpr))]["hash"][0][int(rng.integers(0, 32))] ^= 1
        elif kind == 2: w.validators[0, int(rng.integers(0, V))]["signature"][int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3: w.trusted[0, int(rng.integers(0, V))]["voting_power"] += 1
        else: w.latest[0] = int(w.first_height[0]) + int(rng.integers(1, J * B + 3))
        what.append(kind)
    graphs = int(rng.integers(0, 2))
    _lib.check(_lib.lib().bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(graphs)))
    S = int(w.first_height[0])
    want_rc, want_out, want_res, _ = oracle.header_range(J, B, w.input48(0), w.headers[0], S, int(w.latest[0]), w.validators[0], w.trusted[0])
    circ = CombinedSkipCircuit(V, J, B)
    for _ in range(int(rng.integers(1, 6))):                  # repeated requests of one shape: direct launches, capture, replays
        try:
            out, res, _ = circ.prove(w.input48(0), InputDataFetcher(w.headers[0], S, int(w.latest[0])), w.validators[0], w.trusted[0])
            rc = T.OK
            assert out == want_out, ("H output", J, B, V, n_blocks, what)
            assert clean(res) == clean(want_res), ("H commit", J, B, V, n_blocks, what)
        except _lib.BsxError as e:
            rc = e.status
        assert rc == want_rc, ("H status", J, B, V, n_blocks, graphs, rc, want_rc, what)
    print(f"  H: J={J} B={B} V={V} n_blocks={n_blocks} graphs={graphs} tamper={what} rc={want_rc} ok", flush=True)
    return 1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    t_end, rounds, units = time.time() + budget, 0, [0, 0]
    while time.time() < t_end:
        if rng.integers(0, 3) == 0: units[0] += mode_s(rng)
        else: units[1] += host_tier(rng)
        rounds += 1
    _lib.check(_lib.lib().bsx_set_tuning(_lib.context(0), C.c_uint32(T.TUNE_HOST_GRAPHS), C.c_uint64(0)))
    print(f"soak ok: {rounds} rounds, {units[0]} mode-S commits and {units[1]} host-tier requests compared with the oracle")


if __name__ == "__main__":
    main()
