"""CPU suite: Poseidon over Goldilocks.

Nothing under /root/reference holds a Poseidon constant or vector (the arithmetic lives in plonky2 53c5bc3e, an
un-vendored crate, Cargo.lock:3110-3112), so by the build rules this component is PARITY-UNPINNED BY THE REFERENCE TREE.
It is pinned instead to plonky2's PUBLIC known answers, which three independent implementations here reproduce:
  (1) tools/gen_poseidon_constants.py   Python ints: ChaCha8 generator + definition-level permutation
  (2) oracle/poseidon.c                 C, own ChaCha8 generator, unsigned __int128 `% p` arithmetic
  (3) blobstreamx_amd/csrc/{goldilocks,poseidon}.h   the DEVICE source (limb-split MDS, 32-bit multiply-adds) compiled
      for the host by tests/hostcheck — the exact kernel arithmetic, checked without a GPU
"""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle
from blobstreamx_amd import types as T

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import gen_poseidon_constants as gen  # noqa: E402

P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1

# plonky2's published values [UPSTREAM; recalled from the public source, NOT from /root/reference]:
# poseidon_goldilocks.rs ALL_ROUND_CONSTANTS[0..6) and test_vectors() inputs [0; 12] and [0, 1, .., 11]
PUBLIC_RC_HEAD = [0xb585f766f2144405, 0x7746a55f43921ad7, 0xb2fb0d31cee799b4, 0x0f6760a4803427d7, 0xe10d666650f4e012, 0x8cae14cb07d09bf1]
PUBLIC_KAT = [
    ([0] * 12,
     [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca, 0xd7709673896996dc, 0x46a84e87642f44ed,
      0xd032648251ee0b3c, 0x1c687363b207df62, 0xdf8565563e8045fe, 0x40f5b37ff4254dae, 0xd070f637b431067c, 0x1792b1c4342109d7]),
    (list(range(12)),
     [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2, 0x414754bfebd051f0, 0xcb1f8980294a023f,
      0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056, 0xf045d5eafdc6021f, 0xe4150f77caaa3be5, 0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]),
]
EDGE = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, P - 1, P, P + 1, M64 - 1, M64, 0x8000000000000000, 0xFFFFFFFE00000001]


@pytest.fixture(scope="module")
def hc():
    d = os.path.join(HERE, "hostcheck")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(d, "libhostcheck.so"))
    for f in ("hc_gl_add", "hc_gl_add_canon", "hc_gl_sub", "hc_gl_mul", "hc_gl_pow7", "hc_gl_reduce128", "hc_gl_canonical"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [C.c_uint64] * (1 if f in ("hc_gl_pow7", "hc_gl_canonical") else 2)
    lib.hc_poseidon_rc.restype = C.POINTER(C.c_uint64)
    return lib


def test_round_constants_three_generators_and_public_head():
    py = gen.round_constants()
    assert py[:6] == PUBLIC_RC_HEAD
    assert all(c < P for c in py) and len(py) == 360
    assert list(map(int, oracle.poseidon_round_constants())) == py           # oracle's own C ChaCha8


def test_product_constant_table_is_the_generated_one(hc):
    rc = np.ctypeslib.as_array(hc.hc_poseidon_rc(), shape=(394,))
    py = gen.round_constants()
    assert list(map(int, rc[:360])) == py
    # the committed header is exactly what the generator writes (no hand edits)
    hdr = open(os.path.join(os.path.dirname(HERE), "blobstreamx_amd", "csrc", "poseidon_consts.h")).read()
    assert all(f"0x{c:016x}ull" in hdr for c in py)
    # the folded partial-round constants (round 5): the table the device code indexes behind the 360, and the algebra behind it
    f, g = gen.folded_partial_constants(py)
    assert list(map(int, rc[360:])) == f + g and len(f) == 22 and len(g) == 12
    rng = random.Random(5)
    for st in [[0] * 12, [P - 1] * 12] + [[rng.getrandbits(64) % P for _ in range(12)] for _ in range(20)]:
        assert gen.permute_folded(st, py, f, g) == gen.permute(st, py)


def test_generated_sbox_asm_header_is_current(tmp_path):
    """blobstreamx_amd/csrc/goldilocks_sbox_asm.h is what tools/gen_gl_sbox_asm.py writes (no hand edits)."""
    import importlib.util
    root = os.path.dirname(HERE)
    spec = importlib.util.spec_from_file_location("gen_gl_sbox_asm", os.path.join(root, "tools", "gen_gl_sbox_asm.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    lines = []
    chains = [m.chain(k) for k in range(3)]
    for i in range(len(chains[0])):
        for k in range(3):
            lines.append(chains[k][i])
    hdr = open(os.path.join(root, "blobstreamx_amd", "csrc", "goldilocks_sbox_asm.h")).read()
    got = [l.strip()[1:].split("\\n")[0] for l in hdr.split("\n") if l.strip().startswith('"v_')]
    assert got == lines and len(lines) == 222


@pytest.mark.parametrize("inp,want", PUBLIC_KAT)
def test_permutation_public_kats_all_implementations(hc, inp, want):
    assert gen.permute(inp) == want
    assert list(map(int, oracle.poseidon_permute(np.array(inp, np.uint64)))) == want
    s = np.array(inp, np.uint64)
    hc.hc_poseidon_permute(s.ctypes.data_as(C.c_void_p))
    assert list(map(int, s)) == want


def test_goldilocks_device_arithmetic_vs_python_ints(hc):
    rng = random.Random(7)
    vals = EDGE + [rng.getrandbits(64) for _ in range(300)] + [rng.getrandbits(32) for _ in range(50)]
    for a in vals:
        assert hc.hc_gl_canonical(a) == (a if a < P else a - P)
        assert hc.hc_gl_pow7(a) % P == pow(a, 7, P)
        for b in EDGE + [rng.getrandbits(64) for _ in range(12)]:
            assert hc.hc_gl_add(a, b) % P == (a + b) % P
            assert hc.hc_gl_sub(a, b) % P == (a - b) % P
            assert hc.hc_gl_mul(a, b) % P == (a * b) % P
            assert hc.hc_gl_reduce128(a, b) % P == (a + (b << 64)) % P
            if b < P:
                assert hc.hc_gl_add_canon(a, b) % P == (a + b) % P


def test_mds_layer_vs_definition(hc):
    rng = random.Random(8)
    cases = [[M64] * 12, [P - 1] * 12, [0] * 12, [1 << 63] * 12] + [[rng.getrandbits(64) for _ in range(12)] for _ in range(200)]
    for st in cases:
        s = np.array(st, np.uint64)
        hc.hc_poseidon_mds(s.ctypes.data_as(C.c_void_p))
        want = [(sum(st[(i + k) % 12] * gen.MDS_CIRC[i] for i in range(12)) + st[k] * gen.MDS_DIAG[k]) % P for k in range(12)]
        assert [int(x) % P for x in s] == want


def test_limb_recombination_with_a_signed_low_limb(hc):
    """Round 5: between the partial rounds the lowest limb of a lane may be slightly negative (no borrow from the next limb).  The
    hand-over back to words (poseidon_recombine_signed) against Python integers: bounds of the MDS outputs, the corner where the
    middle limb is zero and the carry makes it negative, random values."""
    hc.hc_poseidon_recombine_signed.restype = C.c_uint64
    hc.hc_poseidon_recombine_signed.argtypes = [C.c_int32, C.c_uint32, C.c_uint32]
    rng = random.Random(12)
    A0MAX, A1MAX = 264 * (1 << 22), 264 * ((1 << 22) + (265 << 10))      # what an MDS layer can put out (poseidon.h invariants): < 2^31
    cases = [(0, 0, 0), (-1, 1, 0), (-1, 0, 1), (-67840, 0, 1), (-67840, 1024 * 264, 0), (-(1 << 20), 0, (1 << 29) - 1), (A0MAX, A1MAX, (1 << 29) - 1),
             (-1, (1 << 22), 0), (-(1 << 22), 1, 0), (-(1 << 22) - 1, 2, 0), (0x3fffff, 0x3fffff, 0xfffff), (-5, 0, 0xfffff + 1)]
    cases += [(rng.randrange(-(1 << 20), A0MAX + 1), rng.randrange(0, A1MAX + 1), rng.randrange(0, 1 << 29)) for _ in range(3000)]
    for a0, a1, a2 in cases:
        v = a0 + (a1 << 22) + (a2 << 44)
        if v < 0:
            continue                                         # a lane's value is never negative (poseidon.h): not a reachable input
        assert int(hc.hc_poseidon_recombine_signed(a0, a1, a2)) == v % P, (a0, a1, a2)


def test_permutation_device_source_vs_oracle_random(hc):
    rng = np.random.default_rng(9)
    for t in range(4000):                 # enough to hit the rare borrow path of the limb-form partial rounds (~3 % of states)
        st = rng.integers(0, 1 << 64, 12, dtype=np.uint64) if t % 3 else np.array([EDGE[int(k)] for k in rng.integers(0, len(EDGE), 12)], np.uint64)
        a = st.copy()
        hc.hc_poseidon_permute(a.ctypes.data_as(C.c_void_p))
        b = oracle.poseidon_permute(st)
        assert (a == b).all(), t
        if t < 8:
            assert list(map(int, b)) == gen.permute([int(x) for x in st])


def test_sponge_and_two_to_one_vs_oracle_and_definition(hc):
    rng = np.random.default_rng(10)
    for n in list(range(1, 20)) + [64, 135, 136, 137]:
        x = rng.integers(0, 1 << 64, n, dtype=np.uint64)
        for noop in (0, 1):
            out = np.zeros(4, np.uint64)
            hc.hc_poseidon_hash(x.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_int(noop), out.ctypes.data_as(C.c_void_p))
            want = oracle.poseidon_hash_or_noop(x) if noop else oracle.poseidon_hash_no_pad(x)
            assert (out == want).all(), (n, noop)
        # definition: overwrite-mode sponge, rate 8
        st = [0] * 12
        for k in range(0, n, 8):
            chunk = [int(v) % P for v in x[k:k + 8]]
            st[:len(chunk)] = chunk
            st = gen.permute(st)
        assert list(map(int, oracle.poseidon_hash_no_pad(x))) == st[:4]
    l, r = rng.integers(0, P, 4, dtype=np.uint64), rng.integers(0, P, 4, dtype=np.uint64)
    out = np.zeros(4, np.uint64)
    hc.hc_poseidon_two_to_one(l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert (out == oracle.poseidon_two_to_one(l, r)).all()
    assert list(map(int, out)) == gen.permute([int(v) for v in l] + [int(v) for v in r] + [0] * 4)[:4]
    small = rng.integers(0, P, 3, dtype=np.uint64)
    assert list(map(int, oracle.poseidon_hash_or_noop(small))) == [int(v) for v in small] + [0]


def test_oracle_merkle_tree_shape_and_cap():
    rng = np.random.default_rng(11)
    el = rng.integers(0, 1 << 32, 1000, dtype=np.uint64)
    tree, cap = oracle.poseidon_merkle_tree(el, 7, 256, 2)          # 143 real rows, 113 zero rows
    assert tree.shape == (2 * 256 - 4, 4) and cap.shape == (4, 4)
    assert (tree[5] == oracle.poseidon_hash_or_noop(el[35:42])).all()
    last = np.zeros(7, np.uint64); last[:6] = el[994:]
    assert (tree[142] == oracle.poseidon_hash_or_noop(last)).all()
    assert (tree[200] == oracle.poseidon_hash_or_noop(np.zeros(7, np.uint64))).all()
    assert (tree[256] == oracle.poseidon_two_to_one(tree[0], tree[1])).all()
    # cap node 0 = root of the first quarter
    lvl = tree[:64]
    while len(lvl) > 1:
        lvl = np.array([oracle.poseidon_two_to_one(lvl[2 * i], lvl[2 * i + 1]) for i in range(len(lvl) // 2)])
    assert (cap[0] == lvl[0]).all()


def test_cpu_baseline_permutation_equals_the_definition():
    """oracle/poseidon.c holds a second, CPU-shaped permutation (128-bit accumulation, no `%`) that only bench.py's
    fused_commitment.cpu_baseline uses: equal to the definition-level one on the public KAT inputs, edge words and random states,
    and its threaded caps driver equals orc_poseidon_merkle_tree over the expanded witness."""
    rng = np.random.default_rng(11)
    P = 0xFFFFFFFF00000001
    states = [np.zeros(12, np.uint64), np.arange(12, dtype=np.uint64), np.full(12, P - 1, np.uint64), np.full(12, 2 ** 64 - 1, np.uint64)]
    states += [rng.integers(0, 2 ** 64, 12, dtype=np.uint64) for _ in range(200)]
    for s in states:
        assert (oracle.poseidon_permute_fast(s) == oracle.poseidon_permute(s)).all()
    lay = T.map_layout(2)
    compact = rng.integers(0, 256, 3 * int(lay["compact_stride"]), dtype=np.uint8)
    bo = int(lay["off_bools"])
    for j in range(3):                       # bools are 0/1, words any u32
        c = compact[j * int(lay["compact_stride"]):(j + 1) * int(lay["compact_stride"])]
        c[bo:] &= 1
    caps = oracle.bench_witness_caps(lay, compact, 3, 135, 128, 2, n_threads=2, reps=2)
    wit = oracle.expand_witness(lay, 3, compact)
    nel = int(lay["n_elements"])
    for j in range(3):
        _, cap = oracle.poseidon_merkle_tree(wit[j * nel:(j + 1) * nel], 135, 128, 2)
        assert (caps[j] == cap).all(), j
