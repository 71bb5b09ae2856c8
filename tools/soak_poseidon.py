#!/usr/bin/env python3
"""Randomised soak of the Poseidon / Goldilocks path against the oracle (a tool beside the suite, `python tools/soak_poseidon.py
120`): permutations of random and edge states (non-canonical words >= p, 0, p - 1, 2^64 - 1), hash_n_to_hash_no_pad of random
lengths, two_to_one, Merkle trees of random element counts / leaf lengths / cap heights, and the pipeline's BSX_PIPE_CAPS mode
(caps straight from the compact bytes, random shape / leaf length / cap height) against the oracle's tree over the oracle's own
expanded witness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, synth
from blobstreamx_amd import types as T
from blobstreamx_amd.poseidon import PoseidonHash, MerkleTree

P = 0xFFFFFFFF00000001
EDGE = np.array([0, 1, P - 1, P, P + 1, (1 << 64) - 1, 0xFFFFFFFF, 1 << 32, (1 << 63)], np.uint64)


def rand_words(rng, shape):
    x = rng.integers(0, 1 << 64, shape, dtype=np.uint64)
    m = rng.random(shape) < 0.15
    x[m] = rng.choice(EDGE, int(m.sum()))
    return x


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    t_end, rounds, n = time.time() + budget, 0, [0, 0, 0, 0]
    H = PoseidonHash()
    while time.time() < t_end:
        kind = int(rng.integers(0, 4))
        if kind == 0:
            s = rand_words(rng, (int(rng.integers(1, 400)), 12))
            got = H.permute(s)
            for i in range(0, s.shape[0], 5):
                want = oracle.poseidon_permute(s[i] % np.uint64(P)) % np.uint64(P)
                assert (got[i] % np.uint64(P) == want).all(), ("permute", s[i])
            n[0] += s.shape[0]
        elif kind == 1:
            length = int(rng.integers(0, 300)); rows = int(rng.integers(1, 200))
            x = rand_words(rng, (rows, length)) if length else np.zeros((rows, 0), np.uint64)
            got = H.hash_no_pad(x)
            for i in range(0, rows, 9):
                assert (got[i] == oracle.poseidon_hash_no_pad(x[i])).all(), ("hash_no_pad", length, i)
            l, r = rand_words(rng, (50, 4)), rand_words(rng, (50, 4))
            t = H.two_to_one(l, r)
            for i in range(0, 50, 3):
                assert (t[i] == oracle.poseidon_two_to_one(l[i] % np.uint64(P), r[i] % np.uint64(P))).all(), "two_to_one"
            n[1] += rows
        elif kind == 2:
            ne = int(rng.integers(1, 6000)); leaf_len = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 80, 135])); cap_h = int(rng.integers(0, 5))
            el = rand_words(rng, ne)
            rows = -(-ne // leaf_len)
            t = MerkleTree(el, leaf_len, cap_h) if (1 << cap_h) <= max(1, rows) else MerkleTree(el, leaf_len, cap_h, n_leaves=1 << cap_h)
            tree, cap = oracle.poseidon_merkle_tree(el, leaf_len, t.n_leaves, cap_h)
            assert (t.digests == tree).all() and (t.cap == cap).all(), ("merkle", ne, leaf_len, cap_h)
            n[2] += 1
        else:
            from blobstreamx_amd.engine import Pipeline
            J = int(rng.choice([1, 2, 4, 8])); B = int(rng.choice([2, 8, 16, 32, 64])); V = int(rng.choice([1, 6, 20])); E = int(rng.choice([1, 2]))
            R = E * int(rng.integers(1, 3)); leaf_len = int(rng.choice([8, 80, 135, 200])); cap_h = int(rng.integers(0, 5))
            w = synth.Workload(int(rng.integers(1, 1 << 20)), R, J, B, v=V, n_blocks=int(rng.integers(1, J * B + 1)))
            p = Pipeline(J, B, V, R, n_chunks=E, with_witness=False, with_caps=True, leaf_len=leaf_len, cap_height=cap_h)
            p.upload_workload(w)
            for _ in range(int(rng.integers(1, 4))): p.step()
            res = p.download()
            nel = int(p.ml["n_elements"])
            n_leaves = int(p.L.bsx_witness_leaf_count(__import__("ctypes").c_uint64(nel), __import__("ctypes").c_uint32(leaf_len)))
            ch = min(cap_h, n_leaves.bit_length() - 1)
            caps = [p.caps_numpy(e) for e in range(E)]
            for r in range(R):
                rc, out, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r],
                                                     want_witness=True)
                assert rc == T.OK and res["output64"][r].tobytes() == out
                full = oracle.expand_range_witness(J, B, cw)
                e, k = divmod(r, p.Rc)
                for j in range(J):
                    _, cap = oracle.poseidon_merkle_tree(full[j * nel:(j + 1) * nel], leaf_len, n_leaves, ch)
                    tree, _ = oracle.poseidon_merkle_tree(full[j * nel:(j + 1) * nel], leaf_len, n_leaves, ch)
                    assert (caps[e][0][k * J + j] == tree).all() and (caps[e][1][k * J + j] == cap).all(), ("caps", J, B, V, R, leaf_len, cap_h, r, j)
            n[3] += R * J
            del p
        rounds += 1
    print(f"soak ok: {rounds} rounds: {n[0]} permutations, {n[1]} sponge rows, {n[2]} Merkle trees, {n[3]} map-job caps from the pipeline compared with the oracle")


if __name__ == "__main__":
    main()
