"""Seeded synthetic INPUT generator (ctypes binding of synth/libsynth.so).

Manufactures inputs only — linked Tendermint header chains, validator sets and signed commits of
the shape SURVEY.md §8(d) defines — for tests, smoke() and bench.py.  It computes none of the hot
path's outputs and is independent of both the product library and the test oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from blobstreamx_amd import types as T

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SYNTH_SO_OVERRIDE") or os.path.join(_DIR, "libsynth.so")   # override: the sanitizer build
_lib = None

BASE_SEED = 0xB10B57
CHAIN_ID = "celestia"
START_HEIGHT = 1_000_000
TIME0 = 1_700_000_000


def build(force=False):
    srcs = [os.path.join(_DIR, "synth.c"), os.path.join(_DIR, "..", "include", "bsx.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-C", _DIR, "-B", "libsynth.so"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None  # data_as keeps `a` alive


class ValidatorSet:
    def __init__(self, seed, v):
        self.v = v
        self.sk_seeds = np.zeros((max(v, 1), 32), np.uint8)
        self.pubkeys = np.zeros((max(v, 1), 32), np.uint8)
        self.powers = np.zeros(max(v, 1), np.uint64)
        self.hash = np.zeros(32, np.uint8)
        rc = lib().synth_validator_set(C.c_uint64(seed), C.c_uint32(v), _p(self.sk_seeds), _p(self.pubkeys),
                                       _p(self.powers), _p(self.hash))
        assert rc == 0

    def rotated(self, seed, n_replace):
        """A copy with `n_replace` slots (chosen by `seed`) re-keyed and re-powered: the next validator set along the chain."""
        import copy
        o = copy.copy(self)
        o.sk_seeds, o.pubkeys, o.powers, o.hash = self.sk_seeds.copy(), self.pubkeys.copy(), self.powers.copy(), np.zeros(32, np.uint8)
        rc = lib().synth_validator_set_rotate(C.c_uint64(seed), C.c_uint32(self.v), C.c_uint32(n_replace), _p(o.sk_seeds), _p(o.pubkeys),
                                              _p(o.powers), _p(o.hash))
        assert rc == 0
        return o

    def as_validators(self, v_max):
        """The set as bsx_validator slots without signatures (the `trusted` argument of header_range)."""
        out = np.zeros(v_max, T.VALIDATOR)
        out["pubkey"][:self.v] = self.pubkeys[:self.v]
        out["voting_power"][:self.v] = self.powers[:self.v]
        out["enabled"][:self.v] = 1
        return out


def chain(seed, n_headers, valset_hash, start_height=START_HEIGHT, chain_id=CHAIN_ID, time0=TIME0):
    """n linked headers -> (headers ndarray[HEADER], hashes[n,32])"""
    headers = np.zeros(n_headers, T.HEADER)
    hashes = np.zeros((n_headers, 32), np.uint8)
    vh = np.ascontiguousarray(valset_hash, np.uint8)
    rc = lib().synth_chain(C.c_uint64(seed), chain_id.encode(), C.c_uint64(start_height), C.c_uint64(n_headers),
                           C.c_uint64(time0), _p(vh), _p(headers), _p(hashes))
    assert rc == 0
    return headers, hashes


def commits(seed, valset, v_max, heights, block_hashes, time_secs, chain_id=CHAIN_ID, absent_permille=0, n_threads=None, round=0,
            nil_permille=0):
    """One signed commit per (height, block_hash) -> ndarray[n_commits, v_max] of VALIDATOR"""
    heights = np.ascontiguousarray(heights, np.uint64).reshape(-1)
    n = heights.size
    bh = np.ascontiguousarray(block_hashes, np.uint8).reshape(n, 32)
    ts = np.ascontiguousarray(time_secs, np.uint64).reshape(n)
    out = np.zeros((n, v_max), T.VALIDATOR)
    if n_threads is None:
        n_threads = os.cpu_count() or 1
    rc = lib().synth_commits(C.c_uint64(seed), chain_id.encode(), C.c_uint32(n), _p(heights), _p(bh), _p(ts),
                             C.c_uint32(valset.v), C.c_uint32(v_max), _p(valset.sk_seeds), _p(valset.pubkeys),
                             _p(valset.powers), C.c_uint32(absent_permille), C.c_int(n_threads), _p(out), C.c_uint64(round),
                             C.c_uint32(nil_permille))
    assert rc == 0
    return out


def sha256(m):
    out = np.zeros(32, np.uint8)
    buf = np.frombuffer(bytes(m), np.uint8).copy() if len(m) else np.zeros(1, np.uint8)
    lib().synth_sha256(_p(buf), C.c_size_t(len(m)), _p(out))
    return out.tobytes()


def sha512(m):
    out = np.zeros(64, np.uint8)
    buf = np.frombuffer(bytes(m), np.uint8).copy() if len(m) else np.zeros(1, np.uint8)
    lib().synth_sha512(_p(buf), C.c_size_t(len(m)), _p(out))
    return out.tobytes()


def ed25519_keypair(seed32):
    pk = np.zeros(32, np.uint8)
    lib().synth_ed25519_keypair(_p(np.frombuffer(bytes(seed32), np.uint8).copy()), _p(pk))
    return pk.tobytes()


def ed25519_sign(seed32, msg):
    sig = np.zeros(64, np.uint8)
    buf = np.frombuffer(bytes(msg), np.uint8).copy() if len(msg) else np.zeros(1, np.uint8)
    lib().synth_ed25519_sign(_p(np.frombuffer(bytes(seed32), np.uint8).copy()), _p(buf), C.c_size_t(len(msg)), _p(sig))
    return sig.tobytes()


class Workload:
    """R independent header_range instances of one shape.

    Range r: trusted block S_r = START_HEIGHT + r * 10_000, target E_r = S_r + n_blocks
    (n_blocks <= J*B), chain_head = S_r + J*B + 2 (so that no hint slot is zero padded when
    n_blocks == J*B; smaller n_blocks leave real headers in disabled slots exactly as the reference's
    hint does, circuits/input.rs:160-165).  headers_per_range = J*B + 1.
    mode "F": one commit per range (the target header's).  mode "S": one commit per header slot
    1..J*B of every range (stress).
    """

    def __init__(self, config_index, n_ranges, nb_map_jobs, batch_size, v, v_max=None, n_blocks=None, mode="F",
                 absent_permille=0, chain_id=CHAIN_ID, round=0, nil_permille=0, rotate_permille=0):
        """rotate_permille = p: range r + 1 is signed by range r's validator set with p / 1000 of its slots re-keyed (at least one when
        p > 0) — validator sets change along a chain (circuits/fetcher.rs:60-87 exists because they do); 1000 = every range its own set.
        The trusted set of a range is its target set (one validators_hash per range's headers)."""
        J, B = nb_map_jobs, batch_size
        self.J, self.B, self.R, self.v = J, B, n_ranges, v
        self.v_max = v_max or v
        self.n_blocks = J * B if n_blocks is None else n_blocks
        self.mode = mode
        seed = BASE_SEED + config_index
        self.seed = seed
        self.hpr = J * B + 1
        self.valset = ValidatorSet(seed, v)
        self.rotate_permille = rotate_permille
        self.valsets = [self.valset]
        for r in range(1, n_ranges):
            n_rep = 0 if not rotate_permille else max(1, (v * rotate_permille + 999) // 1000)
            self.valsets.append(self.valsets[-1].rotated(seed * 7919 + r, n_rep) if n_rep else self.valset)
        self.headers = np.zeros((n_ranges, self.hpr), T.HEADER)
        self.hashes = np.zeros((n_ranges, self.hpr, 32), np.uint8)
        self.ranges = np.zeros(n_ranges, T.SHARED_CTX)
        self.latest = np.zeros(n_ranges, np.uint64)
        self.first_height = np.zeros(n_ranges, np.uint64)
        def gen(r):      # chains are independent: generate them on a thread pool (the C call releases the GIL)
            return chain(seed * 1000003 + r, self.hpr, self.valsets[r].hash, start_height=START_HEIGHT + r * 10_000, chain_id=chain_id)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as pool:
            chains = list(pool.map(gen, range(n_ranges)))
        for r in range(n_ranges):
            S = START_HEIGHT + r * 10_000
            hd, hs = chains[r]
            self.headers[r], self.hashes[r] = hd, hs
            self.first_height[r] = S
            self.latest[r] = S + J * B + 2
            self.ranges[r]["start_block"] = S
            self.ranges[r]["end_block"] = S + self.n_blocks
            self.ranges[r]["start_header_hash"] = hs[0]
            self.ranges[r]["end_header_hash"] = hs[self.n_blocks]
        # commits
        if mode == "F":
            idx = np.full((n_ranges, 1), self.n_blocks)
        else:
            idx = np.tile(np.arange(1, self.hpr), (n_ranges, 1))
        self.commit_header_index = idx
        heights = (self.first_height[:, None] + idx.astype(np.uint64)).reshape(-1)
        bh = np.stack([self.hashes[r, idx[r]] for r in range(n_ranges)]).reshape(-1, 32)
        ts = (TIME0 + 12 * idx).astype(np.uint64).reshape(-1)
        self.commit_hashes = bh.copy()
        if not rotate_permille:
            self.validators = commits(seed ^ 0xC0FFEE, self.valset, self.v_max, heights, bh, ts, chain_id=chain_id,
                                      absent_permille=absent_permille, round=round, nil_permille=nil_permille)
            self.trusted = np.tile(self.valset.as_validators(self.v_max), (n_ranges, 1))
        else:                       # every range signed by ITS set
            per = idx.shape[1]
            parts = [commits((seed ^ 0xC0FFEE) + 31 * r, self.valsets[r], self.v_max, heights[r * per:(r + 1) * per], bh[r * per:(r + 1) * per],
                             ts[r * per:(r + 1) * per], chain_id=chain_id, absent_permille=absent_permille, round=round, nil_permille=nil_permille)
                     for r in range(n_ranges)]
            self.validators = np.concatenate(parts, axis=0)
            self.trusted = np.stack([self.valsets[r].as_validators(self.v_max) for r in range(n_ranges)])

    def input48(self, r):
        rg = self.ranges[r]
        return (int(rg["start_block"]).to_bytes(8, "big") + bytes(rg["start_header_hash"]) +
                int(rg["end_block"]).to_bytes(8, "big"))
