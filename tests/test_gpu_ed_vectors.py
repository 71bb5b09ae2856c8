"""GPU suite: third-party Ed25519 / SHA-512 vectors THROUGH THE HIP PATH: RFC 8032 §7.1 TEST 1-3 and
SHA(abc), plus constructed edge vectors (non-canonical s, small-order A, mixed-order R, y >= p, x = 0 with the sign bit, off
curve) whose expected verdicts are written down in tests/golden/ed25519_vectors.json with their source (RFC 8032 §5.1.3 /
§5.1.7, cofactorless; generator tests/golden/gen_ed25519_vectors.py, stdlib big integers — no code shared with oracle/ or
the kernels).  The reference's verification site is circuits/header_range.rs:42-48 (builder.skip, body UPSTREAM tendermintx
v1.0.0; host twin circuits/fetcher.rs:76-80): nothing under /root/reference pins these -> `parity_unpinned_by_reference`.

Every kernel form of the signature check sees every vector: the host tier's small fixed-key form (< 8 commits), the 4-lanes-
per-signature fixed-key kernel with batch inversion (>= 8 commits), the one-lane-per-signature by-key layout (>= 32 commits),
slots whose key differs from their table row (generic fall-back inside the keyed kernel), the generic kernel itself, and the
latency form (R decoded ahead, 16 / 8 lanes per signature, projective comparison: bsx_dev_ed25519_decode_r + _verify_keyed_r) that
the host tier takes below 65,536 signatures (16 lanes up to 8,192 signatures: 300 commits x 28 slots = 8,400 take 8)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from blobstreamx_amd import _lib
from blobstreamx_amd import types as T
from blobstreamx_amd.builder import verify_commits

pytestmark = pytest.mark.gpu

VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ed25519_vectors.json")))
ALL = VEC["rfc8032"] + VEC["edge"]


def _slots(vectors):
    v = np.zeros(len(vectors), T.VALIDATOR)
    for i, e in enumerate(vectors):
        pk, m, sig = (bytes.fromhex(e[k]) for k in ("public_key", "message", "signature"))
        assert len(m) <= T.VALIDATOR_MSG_MAX
        v[i]["pubkey"] = np.frombuffer(pk, np.uint8)
        v[i]["signature"] = np.frombuffer(sig, np.uint8)
        v[i]["message"][:len(m)] = np.frombuffer(m, np.uint8)
        v[i]["message_len"] = len(m)
        v[i]["voting_power"] = 10 + i
        v[i]["enabled"] = 1
        v[i]["is_signed"] = 1
    return v


WANT = np.array([e["valid"] for e in ALL], np.uint8)


@pytest.mark.parametrize("n_commits", [1, 3, 8, 40, 300])
def test_public_vectors_through_verify_commits_parity_unpinned_by_reference(n_commits):
    """bsx_verify_commits (host tier): commit c holds the vectors rotated by c slots, so from the second commit on every slot's
    key differs from its table row (built from commit 0) — the keyed kernel's generic fall-back judges them — while commit 0 is
    judged through the fixed-key tables."""
    base = _slots(ALL)
    V = base.size
    vals = np.stack([np.roll(base, c) for c in range(n_commits)])
    res, ok = verify_commits(vals, np.zeros((n_commits, 32), np.uint8))
    for c in range(n_commits):
        assert (ok[c] == np.roll(WANT, c)).all(), (c, [ALL[(i - c) % V]["name"] for i in np.nonzero(ok[c] != np.roll(WANT, c))[0]])
        assert res[c]["n_signed"] == V and res[c]["n_bad_signature"] == V - WANT.sum()
        assert res[c]["n_bad_message"] == V              # none of these messages is a vote for the (zero) header hash
        assert res[c]["signed_power"] == 0


@pytest.mark.parametrize("n_commits", [1, 9, 33, 300, 2400])
def test_public_vectors_same_keys_every_commit_parity_unpinned_by_reference(n_commits):
    """The fixed-key tables judge EVERY slot (all commits carry the vectors in the same order): small form, 4-lane form with
    batch inversion, by-key layout."""
    base = _slots(ALL)
    vals = np.stack([base] * n_commits)
    res, ok = verify_commits(vals, np.zeros((n_commits, 32), np.uint8))
    assert (ok == WANT[None, :]).all(), [ALL[i]["name"] for i in np.nonzero((ok != WANT[None, :]).any(axis=0))[0]]


def test_public_vectors_device_tier_parity_unpinned_by_reference():
    """Device tier: the SHA-512 challenge kernel's raw digests against hashlib and the RFC's SHA(abc) answer, the reduced
    scalars against Python integers, then the GENERIC signature kernel (no tables) on those challenges."""
    import torch
    L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493
    base = _slots(ALL)
    n = base.size
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    dev = torch.device("cuda:0")
    dv = torch.from_numpy(base.view(np.uint8).reshape(-1).copy()).to(dev)
    dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    dd = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), dp(dd)))
    _lib.check(L.bsx_dev_ed25519_verify(ctx, st, dp(dv), dp(dh), C.c_uint64(n), dp(dok)))
    torch.cuda.synchronize()
    dig, h = dd.cpu().numpy().reshape(n, 64), dh.cpu().numpy().reshape(n, 32)
    for i, e in enumerate(ALL):
        pk, m, sig = (bytes.fromhex(e[k]) for k in ("public_key", "message", "signature"))
        want = hashlib.sha512(sig[:32] + pk + m).digest()
        assert dig[i].tobytes() == want, e["name"]
        if "sha512_challenge" in e:
            assert want.hex() == e["sha512_challenge"]
        assert int.from_bytes(h[i].tobytes(), "little") == int.from_bytes(want, "little") % L_ORDER, e["name"]
    assert (dok.cpu().numpy() == WANT).all(), [ALL[i]["name"] for i in np.nonzero(dok.cpu().numpy() != WANT)[0]]
    # FIPS 180-4: SHA-512("abc") is the message of RFC 8032's TEST SHA(abc)
    assert hashlib.sha512(b"abc").hexdigest() == VEC["rfc8032"][3]["message"]


def test_latency_form_device_tier_parity_unpinned_by_reference():
    """bsx_dev_ed25519_decode_r + bsx_dev_ed25519_verify_keyed_r directly (what the pipeline's commit check runs): every vector,
    at sizes on both sides of the 16 / 8 lanes-per-signature switch, against the expected verdicts AND against the
    inversion-based form (bsx_dev_ed25519_verify_keyed) on the same inputs."""
    import torch
    base = _slots(ALL)
    V = base.size
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tab = torch.zeros(int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V))), dtype=torch.uint8, device=dev)
    for n_commits in (1, 5, 293, 400):
        vals = np.stack([base] * n_commits)
        vals[n_commits // 2] = np.roll(base, 3)                      # one commit whose keys differ from the table rows
        n = n_commits * V
        dv = torch.from_numpy(vals.view(np.uint8).reshape(-1).copy()).to(dev)
        dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        rd = torch.zeros(int(L.bsx_ed25519_decoded_r_bytes(C.c_uint64(n))), dtype=torch.uint8, device=dev)
        ok_new = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ok_old = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
        _lib.check(L.bsx_dev_ed25519_decode_r(ctx, st, dp(dv), C.c_uint64(n), dp(rd)))
        _lib.check(L.bsx_dev_ed25519_keytable(ctx, st, dp(dv), C.c_uint32(V), dp(tab)))
        _lib.check(L.bsx_dev_ed25519_verify_keyed_r(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(rd), dp(ok_new)))
        _lib.check(L.bsx_dev_ed25519_verify_keyed(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(tab), C.c_uint32(V), dp(ok_old), None))
        torch.cuda.synchronize()
        want = np.stack([WANT] * n_commits)
        want[n_commits // 2] = np.roll(WANT, 3)
        got = ok_new.cpu().numpy().reshape(n_commits, V)
        assert (got == want).all(), (n_commits, np.argwhere(got != want)[:5])
        assert torch.equal(ok_new, ok_old), n_commits


def test_wide_key_tables_device_tier_parity_unpinned_by_reference():
    """Round 5: key tables with 16-bit digits (bsx_ed25519_keytable_bytes_w / bsx_dev_ed25519_keytable_w / bsx_dev_ed25519_verify_keyed_w,
    BSX_KEYTABLE_BITS_WIDE) judge every public vector as the 12-bit tables do — one lane per signature with and without the
    batch-inversion scratch, four lanes, the small form — and a table is never read in the other width: the wrong promise defers
    every slot to the generic kernel (same verdicts), it does not mis-verify."""
    import torch
    base = _slots(ALL)
    V = base.size
    L, ctx, dp = _lib.lib(), _lib.context(0), _lib.dp
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert int(L.bsx_ed25519_keytable_bytes_w(C.c_uint32(V), C.c_uint32(13))) == 0          # unsupported width
    assert int(L.bsx_ed25519_keytable_bytes_w(C.c_uint32(V), C.c_uint32(12))) == int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V)))
    wide = torch.zeros(int(L.bsx_ed25519_keytable_bytes_w(C.c_uint32(V), C.c_uint32(16))), dtype=torch.uint8, device=dev)
    assert wide.numel() > 10 * int(L.bsx_ed25519_keytable_bytes(C.c_uint32(V)))
    for n_commits in (1, 3, 40, 5200):                       # small form; four lanes; lanes by key; one lane per signature (> 300,000)
        vals = np.stack([base] * n_commits)
        vals[n_commits // 2] = np.roll(base, 5)                      # one commit whose keys differ from the table rows
        n = n_commits * V
        dv = torch.from_numpy(vals.view(np.uint8).reshape(-1).copy()).to(dev)
        dh = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
        scr = torch.zeros(int(L.bsx_ed25519_verify_scratch_bytes(C.c_uint64(n))), dtype=torch.uint8, device=dev)
        _lib.check(L.bsx_dev_sha512_challenge(ctx, st, dp(dv), C.c_uint64(n), dp(dh), None))
        _lib.check(L.bsx_dev_ed25519_keytable_w(ctx, st, dp(dv), C.c_uint32(V), dp(wide), C.c_uint32(16)))
        want = np.stack([WANT] * n_commits)
        want[n_commits // 2] = np.roll(WANT, 5)
        for bits, scratch in ((16, None), (16, scr), (12, scr)):   # the last: a 12-bit kernel on the 16-bit table = every slot deferred
            ok = torch.full((n,), 9, dtype=torch.uint8, device=dev)
            _lib.check(L.bsx_dev_ed25519_verify_keyed_w(ctx, st, dp(dv), dp(dh), C.c_uint64(n), C.c_uint32(V), dp(wide), C.c_uint32(V), dp(ok),
                                                        dp(scratch) if scratch is not None else None, C.c_uint32(bits)))
            torch.cuda.synchronize()
            got = ok.cpu().numpy().reshape(n_commits, V)
            assert (got == want).all(), (n_commits, bits, scratch is not None, np.argwhere(got != want)[:5])
        if n_commits > 1000:
            break
    assert L.bsx_dev_ed25519_keytable_w(ctx, st, dp(dv), C.c_uint32(V), dp(wide), C.c_uint32(14)) == T.ERR_BAD_ARG
