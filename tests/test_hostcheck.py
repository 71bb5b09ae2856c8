"""The device arithmetic headers (blobstreamx_amd/csrc/*.h are __host__ __device__) compiled with g++ and checked
against hashlib, Python integers and the oracle — the exact kernel source, on a machine without a GPU.

tests/hostcheck is a harness; it is never loaded by the product.
"""
import ctypes as C
import hashlib
import os
import random
import subprocess

import pytest

import oracle
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
P25519 = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def hc():
    d = os.path.join(HERE, "hostcheck")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(d, "libhostcheck.so"))
    lib.hc_ed25519_verify.restype = C.c_int
    lib.hc_ed25519_verify_keyed.restype = C.c_int
    lib.hc_ed25519_verify_keyed_wide.restype = C.c_int
    lib.hc_sc_is_canonical.restype = C.c_int
    return lib


def _tm_leaf(x):
    return hashlib.sha256(b"\x00" + x).digest()


def test_leaf_and_inner_hashes_match_hashlib(hc):
    rng = random.Random(1)
    out = C.create_string_buffer(32)
    for n in list(range(0, 55)) + [55, 63, 64, 70, 76]:
        data = bytes(rng.randrange(256) for _ in range(n))
        hc.hc_leaf_hash_var(data, n, out)
        assert out.raw == _tm_leaf(data), n
    for _ in range(8):
        d34, d72, d64 = (bytes(rng.randrange(256) for _ in range(k)) for k in (34, 72, 64))
        hc.hc_leaf_hash_34(d34, out)
        assert out.raw == _tm_leaf(d34)
        hc.hc_leaf_hash_72(d72, out)
        assert out.raw == _tm_leaf(d72)
        hc.hc_leaf_hash_tuple(d64, out)
        assert out.raw == _tm_leaf(d64)
        hc.hc_inner_hash(d64[:32], d64[32:], out)
        assert out.raw == hashlib.sha256(b"\x01" + d64).digest()


def test_sha512_challenge_and_scalar_reduction(hc):
    rng = random.Random(2)
    o64, o32 = C.create_string_buffer(64), C.create_string_buffer(32)
    for n in (0, 1, 47, 48, 63, 64, 100, 111, 112, 124):
        r, a = (bytes(rng.randrange(256) for _ in range(32)) for _ in range(2))
        m = bytes(rng.randrange(256) for _ in range(n))
        hc.hc_sha512_ram(r, a, m, n, o64)
        assert o64.raw == hashlib.sha512(r + a + m).digest(), n
    edge = [0, 1, L - 1, L, L + 1, 2**252, 2**512 - 1, (L << 256) - 1, L * L]
    for x in edge + [rng.getrandbits(512) for _ in range(200)]:
        hc.hc_sc_reduce64((x % 2**512).to_bytes(64, "little"), o32)
        assert int.from_bytes(o32.raw, "little") == (x % 2**512) % L
    for x in (0, 1, L - 1, L, L + 1, 2**256 - 1):
        assert hc.hc_sc_is_canonical(x.to_bytes(32, "little")) == int(x < L)


def test_field_arithmetic(hc):
    rng = random.Random(3)
    o = C.create_string_buffer(32)
    vals = [0, 1, 2, 19, P25519 - 1, P25519 - 19, 2**255 - 20, 2**254, 2**255 - 1] + [rng.getrandbits(255) for _ in range(100)]
    for a in vals:
        ab = a.to_bytes(32, "little")
        for b in vals[:12]:
            hc.hc_fe_mul(ab, b.to_bytes(32, "little"), o)
            assert int.from_bytes(o.raw, "little") == a * b % P25519
        hc.hc_fe_sq(ab, 0, o)
        assert int.from_bytes(o.raw, "little") == a * a % P25519
        hc.hc_fe_sq(ab, 1, o)
        assert int.from_bytes(o.raw, "little") == 2 * a * a % P25519
        hc.hc_fe_invert(ab, o)
        assert int.from_bytes(o.raw, "little") == pow(a, P25519 - 2, P25519)


def _cases(golden):
    """(pk, sig, h, note) — fixture signatures, synthetic ones, and corruptions of each part."""
    out = []
    for blk in golden["blocks"].values():
        for c in blk["commit"]["signatures"]:
            if not c.get("signature"):
                continue
            pk = bytes.fromhex(blk["validators"][c["validator_index"]]["pubkey"])
            out.append((pk, bytes.fromhex(c["signature"]), bytes.fromhex(c["sign_bytes"]), "fixture"))
    rng = random.Random(4)
    for i in range(12):
        seed = bytes(rng.randrange(256) for _ in range(32))
        msg = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 120)))
        out.append((synth.ed25519_keypair(seed), synth.ed25519_sign(seed, msg), msg, "synth"))
    cases = []
    for pk, sig, msg, note in out:
        h = (int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L).to_bytes(32, "little")
        cases.append((pk, sig, h, note))
        flip = lambda b, i: b[:i] + bytes([b[i] ^ 1]) + b[i + 1:]
        cases.append((pk, flip(sig, 3), h, note + "/badR"))
        cases.append((pk, flip(sig, 40), h, note + "/badS"))
        cases.append((pk, sig, flip(h, 9), note + "/badH"))
        cases.append((flip(pk, 5), sig, h, note + "/badA"))
        s_plus_l = (int.from_bytes(sig[32:], "little") + L).to_bytes(32, "little")
        cases.append((pk, sig[:32] + s_plus_l, h, note + "/s+L"))
    return cases


def test_ed25519_generic_and_fixed_key_paths_match_the_oracle(hc, golden):
    cases = _cases(golden)
    assert sum(1 for c in cases if c[3] == "fixture") >= 10
    n_ok = 0
    for pk, sig, h, note in cases:
        want = int(bool(oracle.ed25519_verify_h(pk, sig, h)))
        assert hc.hc_ed25519_verify(pk, sig, h) == want, note
        assert hc.hc_ed25519_verify_keyed(pk, sig, h) == want, note
        if "/" not in note:
            assert want == 1, note
        n_ok += want
    assert 0 < n_ok < len(cases)


# -- tiny Python Edwards arithmetic to forge signatures for CHOSEN (h, s): R = [(s - h a) mod L] B
_D = -121665 * pow(121666, P25519 - 2, P25519) % P25519
_BY = 4 * pow(5, P25519 - 2, P25519) % P25519


def _recover_x(y, sign):
    x2 = (y * y - 1) * pow(_D * y * y + 1, P25519 - 2, P25519) % P25519
    x = pow(x2, (P25519 + 3) // 8, P25519)
    if (x * x - x2) % P25519:
        x = x * pow(2, (P25519 - 1) // 4, P25519) % P25519
    return P25519 - x if (x & 1) != sign else x


def _add(p, q):
    (x1, y1), (x2, y2) = p, q
    k = _D * x1 * x2 * y1 * y2 % P25519
    x3 = (x1 * y2 + x2 * y1) * pow(1 + k, P25519 - 2, P25519) % P25519
    y3 = (y1 * y2 + x1 * x2) * pow(1 - k, P25519 - 2, P25519) % P25519
    return x3, y3


def _mul(k, p):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = _add(acc, p)
        p = _add(p, p)
        k >>= 1
    return acc


def _enc(p):
    return (p[1] | ((p[0] & 1) << 255)).to_bytes(32, "little")


def test_fixed_key_path_digit_edges(hc):
    """Valid signatures forged for scalars whose radix-256 recoding hits the extreme digits (-128, 127, 0) in both
    halves, for h and for s."""
    B = (_recover_x(_BY, 0), _BY)
    seed = bytes(range(32))
    pk = synth.ed25519_keypair(seed)
    d = bytearray(hashlib.sha512(seed).digest()[:32])
    d[0] &= 248; d[31] &= 127; d[31] |= 64
    a = int.from_bytes(d, "little")
    assert _enc(_mul(a, B)) == pk
    edges = [0, 1, 2**128 - 1, 2**128, 2**128 + 1, int("80" * 31, 16), int("7f" * 31, 16), int("ff" * 31, 16), L - 1,
             2**252, 2**252 + 2**127, int("0080" * 15, 16), int("7f80" * 15, 16),
             # radix-65536 digits of s (the B table): -32768, 32767, 0 and alternations
             int("8000" * 15, 16), int("7fff" * 15, 16), int("ffff" * 15, 16), int("00008000" * 7, 16), int("7fff8000" * 7, 16)]
    pairs = [(h, s) for h in edges for s in (edges[3], edges[5], edges[8])] + [(edges[1], s) for s in edges]
    for h, s in pairs:
        h, s = h % L, s % L
        R = _enc(_mul((s - h * a) % L, B))
        sig, hb = R + s.to_bytes(32, "little"), h.to_bytes(32, "little")
        assert oracle.ed25519_verify_h(pk, sig, hb)
        assert hc.hc_ed25519_verify(pk, sig, hb) == 1, (hex(h), hex(s))
        assert hc.hc_ed25519_verify_keyed(pk, sig, hb) == 1, (hex(h), hex(s))
        assert hc.hc_ed25519_verify_keyed_wide(pk, sig, hb) == 1, (hex(h), hex(s))      # key table with 16-bit digits (round 5)
        bad = ((h + 1) % L).to_bytes(32, "little")
        assert hc.hc_ed25519_verify_keyed(pk, sig, bad) == 0
        assert hc.hc_ed25519_verify_keyed_wide(pk, sig, bad) == 0


def test_keycache_rows_keyed_by_public_key(hc):
    """csrc/keycache.h (round 5; host-side bookkeeping of the fixed-key Ed25519 tables): random batches of validator sets that drift,
    swap slots and come back — every active slot is mapped to a row that HOLDS its public key (or is deferred when the table is full of
    rows the batch itself needs), rows are unique per key, `dirty` names exactly the rows whose key changed, a batch of one unchanged set is
    the identity map, and rows persist across batches (a key seen in the previous batch costs no rebuild)."""
    import random
    import numpy as np
    from blobstreamx_amd import types as T
    hc.hc_keycache_new.restype = C.c_void_p
    hc.hc_keycache_new.argtypes = [C.c_uint32, C.c_uint32]
    hc.hc_keycache_free.argtypes = [C.c_void_p]
    hc.hc_keycache_assign.restype = C.c_int
    hc.hc_keycache_assign.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    hc.hc_keycache_keys.restype = C.POINTER(C.c_uint8)
    hc.hc_keycache_keys.argtypes = [C.c_void_p]
    hc.hc_keycache_used.restype = C.POINTER(C.c_uint8)
    hc.hc_keycache_used.argtypes = [C.c_void_p]
    rng = random.Random(31)
    for V, N in ((5, 5), (8, 20), (20, 64), (100, 232)):
        k = hc.hc_keycache_new(V, N)
        pool = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(3 * N)]
        base = pool[:V]
        prev_rows = {}
        for batch in range(40):
            R = rng.randrange(1, 7)
            sets = np.zeros((R, V), T.VALIDATOR)
            keys = list(base)
            for r in range(R):
                if batch and rng.random() < 0.5:                   # drift: a few slots re-keyed from the pool
                    for _ in range(rng.randrange(0, 3)):
                        keys[rng.randrange(V)] = rng.choice(pool[:N + V] if batch % 7 else pool)
                if rng.random() < 0.2:                              # two slots swap
                    a, b = rng.randrange(V), rng.randrange(V)
                    keys[a], keys[b] = keys[b], keys[a]
                for i in range(V):
                    sets[r, i]["pubkey"] = np.frombuffer(keys[i], np.uint8)
                    active = rng.random() < 0.9
                    sets[r, i]["enabled"], sets[r, i]["is_signed"] = 1, int(active)
            base = keys
            rows = np.zeros(R * V, np.uint32)
            dirty = np.zeros(N, np.uint32)
            nd, deferred = C.c_uint32(0), C.c_uint64(0)
            ident = hc.hc_keycache_assign(k, sets.ctypes.data_as(C.c_void_p), R, rows.ctypes.data_as(C.c_void_p), dirty.ctypes.data_as(C.c_void_p),
                                          C.byref(nd), C.byref(deferred))
            tab = np.ctypeslib.as_array(hc.hc_keycache_keys(k), shape=(N, 32)).copy()
            used = np.ctypeslib.as_array(hc.hc_keycache_used(k), shape=(N,)).copy()
            rows = rows.reshape(R, V)
            n_def = 0
            for r in range(R):
                for i in range(V):
                    if not sets[r, i]["is_signed"]:
                        assert rows[r, i] == i                                          # inactive slots keep their own index
                        continue
                    q = int(rows[r, i])
                    if q == 0xFFFFFFFF:
                        n_def += 1
                        continue
                    assert q < N and used[q] and tab[q].tobytes() == sets[r, i]["pubkey"].tobytes(), (V, N, batch, r, i, q)
            assert n_def == deferred.value
            distinct_active = {sets[r, i]["pubkey"].tobytes() for r in range(R) for i in range(V) if sets[r, i]["is_signed"]}
            if len(distinct_active) <= N:
                assert n_def == 0, (V, N, batch, len(distinct_active))                # there is room: nothing goes to the generic kernel
            held = [tab[q].tobytes() for q in range(N) if used[q]]
            assert len(held) == len(set(held))                                          # one row per key
            d = set(int(x) for x in dirty[:nd.value])
            now_rows = {tab[q].tobytes(): q for q in range(N) if used[q]}
            changed = {q for key, q in now_rows.items() if prev_rows.get(key) != q}
            assert d == changed, (V, N, batch, sorted(d), sorted(changed))             # dirty = exactly the rows that hold a new key
            assert bool(ident) == bool((rows == np.arange(V)[None, :]).all())
            prev_rows = now_rows
        hc.hc_keycache_free(k)
