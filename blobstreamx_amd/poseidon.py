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
