"""Every launch form of the fixed-key Ed25519 kernel (1 or 4 lanes per signature x signature-major or key-major lane order
x with / without the batch-inversion scratch, and the small-batch decode-R form) against the oracle, on signatures FORGED for
chosen (h, s): scalars whose radix-4096 (h, per-key tables) and radix-65536 (s, table of B) recodings hit the extreme digits, plus
random ones, valid and invalid.  What the vectors protect: the per-validator Ed25519 check of builder.skip
(/root/reference/circuits/header_range.rs:42-48).

The product library chooses the form from the batch size and reads NO environment variable; the forms are forced through the
BSX_ED_* knobs of the EXPERIMENTS build (blobstreamx_amd/lib/libbsx_exp.so, `make EXPERIMENTS=1`, built by
__graft_entry__.build()), loaded through BSX_LIB_OVERRIDE in one subprocess per form (the knobs are read once per process).  The
child reports bsx_debug_last_launch_form(0) after every call and the parent ASSERTS that the requested kernel is the one that ran
(kernels.h BSX_NOTE_FORM): a parametrisation that silently runs the auto-chosen form fails."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493


def _forge():
    """(pk, [(sig64, h32, want)])"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import oracle
    import synth
    from test_hostcheck import _BY, _enc, _mul, _recover_x
    B = (_recover_x(_BY, 0), _BY)
    seed = bytes(range(32))
    pk = synth.ed25519_keypair(seed)
    d = bytearray(hashlib.sha512(seed).digest()[:32])
    d[0] &= 248; d[31] &= 127; d[31] |= 64
    a = int.from_bytes(d, "little")
    edges = [0, 1, 2 ** 128 - 1, 2 ** 128, int("80" * 31, 16), int("7f" * 31, 16), int("ff" * 31, 16), L_ORDER - 1, 2 ** 252,
             int("0080" * 15, 16), int("7f80" * 15, 16), int("8000" * 15, 16), int("7fff" * 15, 16), int("ffff" * 15, 16),
             int("00008000" * 7, 16), int("7fff8000" * 7, 16)]
    rng = np.random.default_rng(5)
    rnd = [int.from_bytes(rng.bytes(32), "little") % L_ORDER for _ in range(12)]
    pairs = [(h, s) for h in edges for s in (edges[4], edges[7], edges[11])] + [(edges[1], s) for s in edges] + list(zip(rnd, rnd[::-1]))
    out = []
    for h, s in pairs:
        h, s = h % L_ORDER, s % L_ORDER
        R = _enc(_mul((s - h * a) % L_ORDER, B))
        sig, hb = R + s.to_bytes(32, "little"), h.to_bytes(32, "little")
        out.append((sig, hb, 1))
        out.append((sig, ((h + 1) % L_ORDER).to_bytes(32, "little"), 0))
        out.append((R + ((s + 1) % L_ORDER).to_bytes(32, "little"), hb, 0))
    out.append((out[0][0][:32] + (int.from_bytes(out[0][0][32:], "little") + L_ORDER).to_bytes(32, "little"), out[0][1], 0))   # s + L
    for sig, hb, want in out:
        assert int(bool(oracle.ed25519_verify_h(pk, sig, hb))) == want
    # adversarial ENCODINGS of R (the small-batch kernel decodes R and compares points, the others encode and compare bytes:
    # the accept sets must coincide).  [s]B - [h]A = identity for s = h a: the true R is (0, 1).
    P = 2 ** 255 - 19
    adv = []
    for h in (1, rnd[0], edges[7]):
        s_ = h * a % L_ORDER
        sb, hb = s_.to_bytes(32, "little"), (h % L_ORDER).to_bytes(32, "little")
        adv.append(((1).to_bytes(32, "little") + sb, hb))                              # identity, canonical: valid
        adv.append(((1 | 1 << 255).to_bytes(32, "little") + sb, hb))                   # x = 0 with the sign bit set
        adv.append(((P + 1).to_bytes(32, "little") + sb, hb))                          # y = p + 1: non-canonical identity
        adv.append(((P - 1).to_bytes(32, "little") + sb, hb))                          # (0, -1): on the curve, another point
    sig0, hb0, _ = out[0]
    r0 = int.from_bytes(sig0[:32], "little")
    adv.append(((r0 ^ (1 << 255)).to_bytes(32, "little") + sig0[32:], hb0))            # -R: same y, other sign
    adv.append(((2).to_bytes(32, "little") + sig0[32:], hb0))                          # y = 2 is not on the curve
    adv.append((((r0 & (2 ** 255 - 1)) + P if (r0 & (2 ** 255 - 1)) < 19 else P + 3).to_bytes(32, "little") + sig0[32:], hb0))   # non-canonical y
    adv.append(((2 ** 255 - 1).to_bytes(32, "little") + sig0[32:], hb0))               # y = 2^255 - 1 >= p
    n_valid = 0
    for sig, hb in adv:
        want = int(bool(oracle.ed25519_verify_h(pk, sig, hb)))
        n_valid += want
        out.append((sig, hb, want))
    assert n_valid == 3                                                                # exactly the canonical identity encodings
    return pk, out


def _child():
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    from blobstreamx_amd import _lib
    from blobstreamx_amd import types as T
    pk, cases = _forge()
    v_max = 3                                    # three slots, all holding the same key; slot 2 of the table is left unbuilt
    n_commits = (len(cases) + v_max - 1) // v_max
    vals = np.zeros(n_commits * v_max, T.VALIDATOR)
    hs = np.zeros((n_commits * v_max, 32), np.uint8)
    want = np.zeros(n_commits * v_max, np.uint8)
    for i, (sig, hb, w) in enumerate(cases):
        vals[i]["pubkey"] = np.frombuffer(pk, np.uint8)
        vals[i]["signature"] = np.frombuffer(sig, np.uint8)
        vals[i]["enabled"], vals[i]["is_signed"] = 1, 1
        hs[i] = np.frombuffer(hb, np.uint8)
        want[i] = w
    n = vals.size
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    dv = torch.from_numpy(vals.view(np.uint8).copy()).cuda()
    dh = torch.from_numpy(hs.reshape(-1).copy()).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_keys = 2                                   # slot 2 has no table row: generic fallback inside the same call
    bits = int(os.environ.get("BSX_TEST_KT_BITS", "12"))      # 16: the wide key tables of round 5 (BSX_KEYTABLE_BITS_WIDE)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes_w(C.c_uint32(n_keys), C.c_uint32(bits))), dtype=torch.uint8, device="cuda")
    _lib.check(L.bsx_dev_ed25519_keytable_w(ctx, st, dp(dv), C.c_uint32(n_keys), dp(tab), C.c_uint32(bits)))
    scr = torch.zeros(int(L.bsx_ed25519_verify_scratch_bytes(C.c_uint64(n))), dtype=torch.uint8, device="cuda")
    L.bsx_debug_last_launch_form.restype = C.c_uint32        # AttributeError here = the product library was loaded, not libbsx_exp.so
    forms = []
    for scratch in (None, scr):
        ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        _lib.check(L.bsx_dev_ed25519_verify_keyed_w(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(v_max), dp(tab), C.c_uint32(n_keys),
                                                    dp(ok), dp(scratch) if scratch is not None else None, C.c_uint32(bits)))
        forms.append(int(L.bsx_debug_last_launch_form(C.c_uint32(0))))
        torch.cuda.synchronize()
        got = ok.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (os.environ.get("BSX_ED_SPLIT"), os.environ.get("BSX_ED_BY_KEY"), scratch is not None, bad[:8], got[bad[:8]])
    print("ok", n, int(want.sum()), n_commits, "forms", *("0x%x" % f for f in forms))


EXP_LIB = os.path.join(ROOT, "blobstreamx_amd", "lib", "libbsx_exp.so")


def _expected_forms(split, by_key, small, n_commits):
    """(form without scratch, form with scratch) the launcher must report — kernels.h BSX_NOTE_FORM, kernels_ed.hip
    bsxk_ed25519_verify_keyed: 0x300 = k_ed25519_verify_keyed_small, 0x400 | SPLIT | BYKEY << 4 | DEFER << 5 = k_ed25519_verify_keyed."""
    if split is None:                            # auto: a batch this small takes four lanes per signature; lanes by key from 32 commits on
        split, by_key = 4, int(n_commits >= 32)
    keyed = lambda scr: 0x400 | split | (0x10 if by_key else 0) | (0x20 if scr else 0)
    no_scratch = 0x300 if (split == 4 and not by_key and small) else keyed(False)
    return no_scratch, keyed(True)


@pytest.mark.gpu
@pytest.mark.parametrize("split,by_key,small,bits", [(1, 0, 1, 12), (1, 1, 1, 12), (4, 0, 1, 12), (4, 0, 0, 12), (4, 1, 1, 12), (None, None, 1, 12),
                                                     (1, 0, 1, 16), (1, 1, 1, 16), (4, 0, 1, 16), (4, 1, 1, 16)])
def test_every_launch_form_on_forged_digit_edges(split, by_key, small, bits):
    """(4, 0, 1) is the small-batch form that decodes R in a second wave and compares projectively instead of encoding.  bits = 16: the
    same forms over key tables with 16-bit digits (the forged scalars hit the extreme radix-65536 digits as well: they are built for s)."""
    assert os.path.exists(EXP_LIB), "libbsx_exp.so is not built: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, BSX_LIB_OVERRIDE=EXP_LIB)
    env.pop("BSX_ED_SPLIT", None); env.pop("BSX_ED_BY_KEY", None)
    env["BSX_ED_SMALL"] = str(small)
    env["BSX_TEST_KT_BITS"] = str(bits)
    if split is not None:
        env["BSX_ED_SPLIT"], env["BSX_ED_BY_KEY"] = str(split), str(by_key)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-1500:]
    words = out.stdout.split()
    n_commits = int(words[3])
    got = tuple(int(x, 16) for x in words[words.index("forms") + 1:])
    assert got == _expected_forms(split, by_key, small, n_commits), (got, _expected_forms(split, by_key, small, n_commits))


def test_the_six_parametrisations_are_five_distinct_kernels_plus_the_auto_choice():
    """The forced forms differ pairwise (per scratch flavour) — the round-4 regression was six runs of ONE kernel."""
    rows = [_expected_forms(s, b, m, 70) for s, b, m in [(1, 0, 1), (1, 1, 1), (4, 0, 1), (4, 0, 0), (4, 1, 1)]]
    assert len({r[0] for r in rows}) == 5 and len({r[1] for r in rows}) == 4       # with a scratch (4,0,1) and (4,0,0) coincide


if __name__ == "__main__" and sys.argv[1:] == ["child"]:
    _child()
