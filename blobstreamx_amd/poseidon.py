"""Host-side mirror of plonky2's Poseidon interface as the header_range path uses it:
`PoseidonGoldilocksConfig` is the hash config of every reference binary (bin/header_range_2048.rs:1-17 through
plonky2x DefaultParameters); plonky2 [UPSTREAM, Cargo.lock:3110-3112] names are kept:

  PoseidonHash.permute / hash_no_pad / hash_or_noop / two_to_one      plonky2 hash/poseidon.rs, hash/hashing.rs
  MerkleTree(leaves, cap_height)                                      plonky2 hash/merkle_tree.rs
  witness_merkle_caps(...)                                            the commitment the prover opens witness-gen with

Every call goes through the C ABI of libbsx.so (include/bsx.h); no arithmetic happens in Python.
"""
import ctypes as C

import numpy as np

from . import _lib

ORDER = 0xFFFFFFFF00000001


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class PoseidonHash:
    def __init__(self, device=0):
        self.device = device

    def _ctx(self):
        return _lib.context(self.device)

    def permute(self, states):
        s = _u64(states).reshape(-1, 12)
        out = np.zeros_like(s)
        _lib.check(_lib.lib().bsx_poseidon_permute(self._ctx(), _lib.p(s), C.c_uint64(s.shape[0]), _lib.p(out)))
        return out

    def hash_no_pad(self, inputs):
        """inputs: [n, len] -> [n, 4] (hash_n_to_hash_no_pad of every row)"""
        x = _u64(inputs)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.zeros((x.shape[0], 4), np.uint64)
        _lib.check(_lib.lib().bsx_poseidon_hash_no_pad(self._ctx(), _lib.p(x) if x.size else None, C.c_uint64(x.shape[0]),
                                                       C.c_uint32(x.shape[1]), _lib.p(out)))
        return out

    def two_to_one(self, left, right):
        l, r = _u64(left).reshape(-1, 4), _u64(right).reshape(-1, 4)
        out = np.zeros_like(l)
        _lib.check(_lib.lib().bsx_poseidon_two_to_one(self._ctx(), _lib.p(l), _lib.p(r), C.c_uint64(l.shape[0]), _lib.p(out)))
        return out


def tree_digests(n_leaves, cap_height):
    return int(_lib.lib().bsx_poseidon_tree_digests(C.c_uint32(n_leaves), C.c_uint32(cap_height)))


def witness_leaf_count(n_elements, leaf_len):
    return int(_lib.lib().bsx_witness_leaf_count(C.c_uint64(n_elements), C.c_uint32(leaf_len)))


class MerkleTree:
    """MerkleTree::<F, PoseidonHash>::new(leaves, cap_height): leaves = rows of `leaf_len` elements of a flat element
    vector, zero padded to n_leaves (a power of two) rows."""

    def __init__(self, elements, leaf_len, cap_height, n_leaves=None, device=0):
        e = _u64(elements).reshape(-1)
        self.leaf_len, self.cap_height = leaf_len, cap_height
        self.n_leaves = n_leaves or witness_leaf_count(e.size, leaf_len)
        nd = tree_digests(self.n_leaves, cap_height)
        self.digests = np.zeros((max(nd, 1), 4), np.uint64)
        _lib.check(_lib.lib().bsx_poseidon_merkle_tree(_lib.context(device), _lib.p(e), C.c_uint64(e.size), C.c_uint32(leaf_len),
                                                       C.c_uint32(self.n_leaves), C.c_uint32(cap_height), _lib.p(self.digests)))

    @property
    def cap(self):
        return self.digests[-(1 << self.cap_height):]


def witness_merkle_caps(layout, witness, n_jobs, leaf_len, cap_height, device=0):
    """One Merkle cap per job over the MATERIALISED witness (host pointer) -> [n_jobs, 2^cap_height, 4]."""
    lay = np.array(layout).reshape(1)
    w = _u64(witness).reshape(-1)
    out = np.zeros((n_jobs, 1 << cap_height, 4), np.uint64)
    _lib.check(_lib.lib().bsx_witness_merkle_caps(_lib.context(device), _lib.p(lay), _lib.p(w), C.c_uint32(n_jobs), C.c_uint32(leaf_len),
                                                  C.c_uint32(cap_height), _lib.p(out)))
    return out


class WitnessCommitter:
    """Device-resident witness-column commitment of n_jobs witnesses of one layout: per job a Poseidon Merkle tree over
    rows of `leaf_len` elements (plonky2 standard_recursion_config has 135 wires per row; cap_height 4), kept in HBM
    as [n_jobs][tree_digests][4] u64.  commit_compact() is the fused path (elements generated on the fly from the
    compact witness — nothing 64x-expanded is ever written); commit_materialised() hashes an expanded witness buffer.
    PyTorch only owns the buffer and the stream."""

    def __init__(self, layout, n_jobs, leaf_len=135, cap_height=4, device=None):
        import torch
        self.lay = np.array(layout).reshape(1)
        self.n_jobs, self.leaf_len = int(n_jobs), int(leaf_len)
        self.nel = int(self.lay["n_elements"][0])
        self.n_leaves = witness_leaf_count(self.nel, leaf_len)
        self.cap_height = min(int(cap_height), self.n_leaves.bit_length() - 1)
        self.nd = tree_digests(self.n_leaves, self.cap_height)
        self.dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.ctx = _lib.context(self.dev.index if self.dev.index is not None else 0)
        self.trees = torch.zeros(max(self.n_jobs * self.nd * 4, 4), dtype=torch.int64, device=self.dev)
        self.n_rows = -(-self.nel // leaf_len)
        # permutations per job: ceil(leaf_len / 8) per real row (the zero rows share one digest) + the tree above
        self.perms_per_job = self.n_rows * (-(-leaf_len // 8) if leaf_len > 4 else 0) + (self.n_leaves - (1 << self.cap_height))

    def _st(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _caps(self):
        _lib.check(_lib.lib().bsx_dev_poseidon_merkle_caps(self.ctx, self._st(), _lib.dp(self.trees), C.c_uint32(self.n_jobs),
                                                           C.c_uint64(4 * self.nd), C.c_uint32(self.n_leaves), C.c_uint32(self.cap_height)))

    def commit_compact(self, d_compact):
        _lib.check(_lib.lib().bsx_dev_witness_leaf_hashes(self.ctx, self._st(), _lib.p(self.lay), C.c_uint32(self.n_jobs), _lib.dp(d_compact),
                                                          C.c_uint32(self.leaf_len), C.c_uint32(self.n_leaves), C.c_uint64(4 * self.nd),
                                                          _lib.dp(self.trees)))
        self._caps()

    def commit_materialised(self, d_witness):
        _lib.check(_lib.lib().bsx_dev_poseidon_leaf_hashes(self.ctx, self._st(), _lib.dp(d_witness), C.c_uint32(self.n_jobs), C.c_uint64(self.nel),
                                                           C.c_uint32(self.leaf_len), C.c_uint32(self.n_leaves), C.c_uint64(4 * self.nd),
                                                           _lib.dp(self.trees)))
        self._caps()

    def caps_numpy(self):
        import torch
        torch.cuda.synchronize(self.dev)
        t = self.trees[:self.n_jobs * self.nd * 4].cpu().numpy().view(np.uint64).reshape(self.n_jobs, self.nd, 4)
        return t[:, self.nd - (1 << self.cap_height):, :]

    def trees_numpy(self):
        import torch
        torch.cuda.synchronize(self.dev)
        return self.trees[:self.n_jobs * self.nd * 4].cpu().numpy().view(np.uint64).reshape(self.n_jobs, self.nd, 4)
