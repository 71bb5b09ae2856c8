"""GPU suite: Poseidon over Goldilocks through the C ABI (bsx_poseidon_*, bsx_dev_*poseidon*, bsx_dev_witness_leaf_hashes)
against the oracle (oracle/poseidon.c) and plonky2's public known answers.  Parity status: see tests/test_oracle_poseidon.py."""
import ctypes as C

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import types as T

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
EDGE = [0, 1, 0xFFFFFFFF, 0x100000000, P - 1, P, P + 1, (1 << 64) - 1, 1 << 63]


def test_permutation_public_kats_on_gpu():
    from blobstreamx_amd.poseidon import PoseidonHash
    from test_oracle_poseidon import PUBLIC_KAT
    H = PoseidonHash()
    out = H.permute(np.array([k[0] for k in PUBLIC_KAT], np.uint64))
    for o, (_, want) in zip(out, PUBLIC_KAT):
        assert list(map(int, o)) == want


def test_permutation_vs_oracle_random_and_edge_states():
    from blobstreamx_amd.poseidon import PoseidonHash
    rng = np.random.default_rng(21)
    st = rng.integers(0, 1 << 64, (5000, 12), dtype=np.uint64)
    st[:64] = np.array([[EDGE[int(k)] for k in rng.integers(0, len(EDGE), 12)] for _ in range(64)], np.uint64)
    got = PoseidonHash().permute(st)
    for i in list(range(200)) + list(range(4900, 5000)):
        assert (got[i] == oracle.poseidon_permute(st[i])).all(), i
    assert (got < np.uint64(P)).all()


@pytest.mark.parametrize("length", [1, 3, 4, 5, 8, 9, 16, 17, 135])
def test_hash_no_pad_and_two_to_one_vs_oracle(length):
    from blobstreamx_amd.poseidon import PoseidonHash
    rng = np.random.default_rng(22 + length)
    x = rng.integers(0, 1 << 64, (300, length), dtype=np.uint64)
    got = PoseidonHash().hash_no_pad(x)
    for i in range(0, 300, 7):
        assert (got[i] == oracle.poseidon_hash_no_pad(x[i])).all(), i
    l, r = rng.integers(0, 1 << 64, (77, 4), dtype=np.uint64), rng.integers(0, 1 << 64, (77, 4), dtype=np.uint64)
    t = PoseidonHash().two_to_one(l, r)
    for i in range(77):
        assert (t[i] == oracle.poseidon_two_to_one(l[i] % np.uint64(P), r[i] % np.uint64(P))).all()


@pytest.mark.parametrize("n,leaf_len,cap_h", [(1000, 7, 2), (4096, 8, 0), (5000, 135, 4), (37, 3, 1), (513, 1, 3)])
def test_merkle_tree_vs_oracle(n, leaf_len, cap_h):
    from blobstreamx_amd.poseidon import MerkleTree
    rng = np.random.default_rng(23)
    el = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    t = MerkleTree(el, leaf_len, cap_h) if (1 << cap_h) <= max(1, -(-n // leaf_len)) else MerkleTree(el, leaf_len, cap_h, n_leaves=1 << cap_h)
    tree, cap = oracle.poseidon_merkle_tree(el, leaf_len, t.n_leaves, cap_h)
    assert (t.digests == tree).all()
    assert (t.cap == cap).all()


@pytest.mark.parametrize("J,B,leaf_len,cap_h", [(2, 8, 135, 4), (4, 32, 135, 4), (2, 64, 80, 3), (2, 4, 8, 0)])
def test_fused_witness_commitment_equals_materialised_and_oracle(J, B, leaf_len, cap_h):
    """The point of §8f row 4: the commitment computed STRAIGHT FROM THE COMPACT BYTES (bsx_dev_witness_leaf_hashes, the
    64x image never exists) equals the one over the materialised Goldilocks witness (bsx_dev_poseidon_leaf_hashes on the
    output of bsx_dev_expand_witness) and the oracle's tree over the oracle's expanded witness — map jobs and reduce
    nodes, every digest of every tree."""
    import torch
    from blobstreamx_amd.engine import HeaderRangeEngine
    V, R = 6, 2
    w = synth.Workload(12, R, J, B, v=V)
    eng = HeaderRangeEngine(J, B, V, R)
    eng.upload_workload(w)
    eng.step()
    torch.cuda.synchronize()
    L, ctx, dp = _lib.lib(), eng.ctx, _lib.dp
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for lay, n_jobs, compact, wit in ((eng.ml, R * J, eng.compact, eng.witness_map), (eng.rl, R * (J - 1), eng.red_compact_local, eng.witness_red_local)):
        nel = int(lay["n_elements"])
        n_leaves = int(L.bsx_witness_leaf_count(C.c_uint64(nel), C.c_uint32(leaf_len)))
        ch = min(cap_h, n_leaves.bit_length() - 1)
        nd = int(L.bsx_poseidon_tree_digests(C.c_uint32(n_leaves), C.c_uint32(ch)))
        lay_a = np.array(lay).reshape(1)
        fused = torch.zeros(n_jobs * nd * 4, dtype=torch.int64, device="cuda")
        mat = torch.zeros_like(fused)
        _lib.check(L.bsx_dev_witness_leaf_hashes(ctx, st, _lib.p(lay_a), C.c_uint32(n_jobs), dp(compact), C.c_uint32(leaf_len),
                                                 C.c_uint32(n_leaves), C.c_uint64(4 * nd), dp(fused)))
        _lib.check(L.bsx_dev_poseidon_merkle_caps(ctx, st, dp(fused), C.c_uint32(n_jobs), C.c_uint64(4 * nd), C.c_uint32(n_leaves), C.c_uint32(ch)))
        _lib.check(L.bsx_dev_poseidon_leaf_hashes(ctx, st, dp(wit), C.c_uint32(n_jobs), C.c_uint64(nel), C.c_uint32(leaf_len),
                                                  C.c_uint32(n_leaves), C.c_uint64(4 * nd), dp(mat)))
        _lib.check(L.bsx_dev_poseidon_merkle_caps(ctx, st, dp(mat), C.c_uint32(n_jobs), C.c_uint64(4 * nd), C.c_uint32(n_leaves), C.c_uint32(ch)))
        torch.cuda.synchronize()
        f = fused.cpu().numpy().view(np.uint64).reshape(n_jobs, nd, 4)
        m = mat.cpu().numpy().view(np.uint64).reshape(n_jobs, nd, 4)
        assert (f == m).all()
        # oracle: its own witness (from its own compact image) hashed by its own Poseidon
        for r in range(R):
            rc, _, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                               w.validators[r], w.trusted[r], want_witness=True)
            assert rc == T.OK
            full = oracle.expand_range_witness(J, B, cw)
            nm = J * int(eng.ml["n_elements"])
            per = J if lay is eng.ml else J - 1
            src = full[:nm] if lay is eng.ml else full[nm:]
            for j in range(per):
                tree, _ = oracle.poseidon_merkle_tree(src[j * nel:(j + 1) * nel], leaf_len, n_leaves, ch)
                assert (f[r * per + j] == tree).all(), (r, j)
    # host tier over the downloaded witness: same caps
    from blobstreamx_amd.poseidon import witness_merkle_caps
    wm, _, _ = eng.witness_numpy()
    nel = int(eng.ml["n_elements"])
    n_leaves = int(L.bsx_witness_leaf_count(C.c_uint64(nel), C.c_uint32(leaf_len)))
    ch = min(cap_h, n_leaves.bit_length() - 1)
    caps = witness_merkle_caps(eng.ml, wm[:2 * nel], 2, leaf_len, ch)
    for j in range(2):
        _, cap = oracle.poseidon_merkle_tree(wm[j * nel:(j + 1) * nel], leaf_len, n_leaves, ch)
        assert (caps[j] == cap).all()


def test_poseidon_argument_errors():
    from blobstreamx_amd.poseidon import MerkleTree
    with pytest.raises(_lib.BsxError):
        MerkleTree(np.arange(100, dtype=np.uint64), 7, 2, n_leaves=12)          # not a power of two
    with pytest.raises(_lib.BsxError):
        MerkleTree(np.arange(100, dtype=np.uint64), 7, 2, n_leaves=8)           # does not cover the elements
    with pytest.raises(_lib.BsxError):
        MerkleTree(np.arange(100, dtype=np.uint64), 7, 9, n_leaves=16)          # cap above the root


def test_merkle_caps_of_different_heights_are_consistent_at_full_size():
    """Size-independent property at the production shape (one 32x64 map job: 449,755 elements, 4096 leaves of 135): the cap
    of height h is the pairwise two_to_one fold of the cap of height h + 1, down to the root, for the fused path — and a
    one-bit change of the compact witness moves exactly one leaf digest and the single cap node above it."""
    import torch
    from blobstreamx_amd.engine import HeaderRangeEngine
    from blobstreamx_amd.poseidon import PoseidonHash, WitnessCommitter
    J, B, V = 32, 64, 4
    w = synth.Workload(13, 1, J, B, v=V)
    eng = HeaderRangeEngine(J, B, V, 1, with_witness=False)
    eng.upload_workload(w)
    eng.step()
    torch.cuda.synchronize()
    caps = {}
    for h in (5, 4, 1, 0):
        wc = WitnessCommitter(eng.ml, J, 135, h)
        wc.commit_compact(eng.compact)
        caps[h] = wc.caps_numpy().copy()
        assert wc.n_leaves == 4096 and caps[h].shape == (J, 1 << h, 4)
    H = PoseidonHash()
    fold = caps[5]
    for h in (4, 3, 2, 1, 0):
        fold = H.two_to_one(fold[:, 0::2].reshape(-1, 4), fold[:, 1::2].reshape(-1, 4)).reshape(J, 1 << h, 4)
        if h in caps:
            assert (fold == caps[h]).all(), h
    wc = WitnessCommitter(eng.ml, J, 135, 4)
    wc.commit_compact(eng.compact)
    before = wc.trees_numpy().copy()
    stride = int(eng.ml["compact_stride"])
    byte = 3 * stride + 20000                                # job 3, bit 160000..160007 -> leaf 160000 // 135 = 1185
    eng.compact[byte] ^= 0x10
    wc.commit_compact(eng.compact)
    after = wc.trees_numpy()
    diff = np.argwhere((before != after).any(axis=2))
    assert set(diff[:, 0]) == {3}
    leaf = (20000 * 8 + 3) // 135
    assert leaf in set(diff[:, 1]) and (diff[:, 1] < 4096).sum() == 1          # exactly one leaf digest changed
    assert (before[3, -16:] != after[3, -16:]).any(axis=1).sum() == 1            # and one of the 16 cap nodes
