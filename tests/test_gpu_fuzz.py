"""GPU suite: seeded differential fuzz — random shapes, global-end positions, chain heads (hint clamp / zero padding)
and random single-byte tampering of headers; the HIP path must agree with the oracle on the status, the assertion
mask, every per-job record and the full witness, pass or fail."""
import os

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import types as T
from blobstreamx_amd.builder import DataCommitmentBuilder, InputDataFetcher

pytestmark = pytest.mark.gpu


def _rec(r):
    r = np.array(r, dtype=T.SUBCHAIN).copy()
    r["_pad"] = 0
    return r.tobytes()


@pytest.mark.parametrize("seed", range(int(os.environ.get("BSX_FUZZ_SEEDS", "12"))))   # BSX_FUZZ_SEEDS=200 for a long soak
def test_prove_data_commitment_fuzz(seed):
    rnd = np.random.default_rng(1000 + seed)
    J = int(rnd.choice([1, 2, 4, 8, 16]))
    B = int(rnd.choice([1, 2, 4, 8, 16, 32]))
    w = synth.Workload(100 + seed, 1, J, B, v=1)
    S = int(w.first_height[0])
    bld = DataCommitmentBuilder()
    for trial in range(6):
        n_blocks = int(rnd.integers(0, J * B + 3))                 # 0 (end == start) .. beyond J*B (A7)
        latest = S + int(rnd.integers(1, J * B + 6))               # chain head anywhere: exercises the latest-2 clamp
        hdrs = w.headers[0].copy()
        if trial >= 3:                                             # tamper one byte of one header
            k = int(rnd.integers(0, hdrs.size))
            raw = hdrs.view(np.uint8).reshape(-1, 512)
            raw[k, int(rnd.integers(16, 512))] ^= 1 << int(rnd.integers(0, 8))
        e_idx = min(n_blocks, J * B)
        ctx = oracle.make_ctx(S, w.hashes[0, 0].tobytes(), S + n_blocks, w.hashes[0, e_idx].tobytes())
        rc_ref, ref = oracle.prove_data_commitment(J, B, ctx, hdrs, S, latest, want_witness=True)
        f = InputDataFetcher(hdrs, S, latest)
        try:
            out = bld.prove_data_commitment(f, J, B, S, w.hashes[0, 0].tobytes(), S + n_blocks, w.hashes[0, e_idx].tobytes(),
                                            want_witness=True, raise_on_assert=False)
            rc = out["rc"]
        except _lib.BsxError as e:
            rc, out = e.status, None
        assert rc == rc_ref, (seed, trial, J, B, n_blocks, latest - S, rc, rc_ref)
        if rc in (T.OK, T.ERR_ASSERT):
            assert out["result"]["assert_fail"] == ref["status"], (seed, trial, hex(out["result"]["assert_fail"]), hex(ref["status"]))
            assert [_rec(r) for r in out["records"]] == [_rec(r) for r in ref["records"]], (seed, trial)
            assert out["data_commitment"] == ref["data_commitment"]
            assert (out["witness"] == oracle.expand_range_witness(J, B, ref["compact"])).all(), (seed, trial)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BSX_ENGINE_FUZZ_SEEDS", "10"))))
def test_engine_fuzz_sharded_shapes(seed):
    """The device-resident pipeline (blobstreamx_amd/engine.py) on random shapes: world size, map jobs, batch size, ranges
    per rank, range length (disabled batches, padded slots through the latest-2 clamp), validator count.  Every rank's
    engine runs on the one GPU with the all-gather emulated; public outputs, statuses and the map-job witness of every
    rank's slice must equal the oracle's."""
    from blobstreamx_amd.engine import HeaderRangeEngine, run_world_on_one_gpu
    rnd = np.random.default_rng(7000 + seed)
    world = int(rnd.choice([1, 2, 4]))
    J = int(rnd.choice([j for j in (2, 4, 8, 16) if j >= world]))
    B = int(rnd.choice([2, 4, 8, 16, 32]))
    R = int(rnd.integers(1, 4))
    V = int(rnd.choice([3, 12, 33]))
    n_blocks = int(rnd.integers(1, J * B + 1))
    w = synth.Workload(300 + seed, R * world, J, B, v=V, n_blocks=n_blocks)
    engs = [HeaderRangeEngine(J, B, V, R, rank=g, world=world) for g in range(world)]
    for e in engs:
        e.upload_workload(w)
    run_world_on_one_gpu(engs)
    ml = T.map_layout(B)
    jc = J // world
    nm = jc * int(ml["n_elements"])
    refs = [oracle.prove_data_commitment(J, B, w.ranges[r:r + 1], w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                         want_witness=True) for r in range(R * world)]
    for g, e in enumerate(engs):
        out = e.download()
        wm, _, _ = e.witness_numpy()
        assert out["header_status"] == 0 and out["assemble_status"] == 0, (seed, g)
        for k in range(R):
            r = g * R + k
            rc, ref_out, _, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                                    w.validators[r], w.trusted[r])
            assert rc == T.OK, (seed, world, J, B, R, V, n_blocks)
            assert out["output64"][k].tobytes() == ref_out, (seed, g, k)
            assert out["range_status"][k] == 0 and out["skip_status"][k] == 0, (seed, g, k)
        for r in range(R * world):
            full = oracle.expand_witness(ml, J, refs[r][1]["compact"])
            assert (wm[r * nm:(r + 1) * nm] == full[g * nm:(g + 1) * nm]).all(), (seed, g, r, world, J, B, n_blocks)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BSX_COMMIT_FUZZ_SEEDS", "10"))))
def test_verify_commits_fuzz(seed):
    """Random validator-set sizes and commit counts around the points where the fixed-key signature kernel changes form
    (signature-major / key-major lane order at 8 and 32 commits; table path from 8 commits on in the host tier), random
    absent / nil / tampered votes and keys that differ from their table row: per-signature verdicts and every commit
    result against the oracle."""
    from blobstreamx_amd.builder import verify_commits
    rng = np.random.default_rng(9000 + seed)
    v_max = int(rng.choice([1, 3, 8, 21, 64, 100, 130]))
    v = int(rng.integers(1, v_max + 1))
    nc = int(rng.choice([1, 2, 7, 8, 9, 31, 32, 33, 70]))
    w = synth.Workload(300 + seed, nc, 1, 2, v=v, v_max=v_max, absent_permille=int(rng.choice([0, 100, 500])),
                       nil_permille=int(rng.choice([0, 200])))
    vals = w.validators.copy()
    hh = w.commit_hashes.copy()
    for _ in range(int(rng.integers(0, 2 * nc + 1))):
        c, k = int(rng.integers(0, nc)), int(rng.integers(0, v_max))
        what = int(rng.integers(0, 4))
        if what == 0:
            vals[c, k]["signature"][int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
        elif what == 1:
            vals[c, k]["pubkey"][int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))      # differs from the table row
        elif what == 2:
            vals[c, k]["message"][int(rng.integers(0, 60))] ^= 1 << int(rng.integers(0, 8))
        else:
            vals[c, k] = vals[int(rng.integers(0, nc)), int(rng.integers(0, v_max))]               # another slot's record
    res, ok = verify_commits(vals, hh)
    for c in range(nc):
        ref, rok = oracle.verify_commit(vals[c], hh[c].tobytes())
        assert (ok[c] == rok).all(), (seed, c, np.nonzero(ok[c] != rok))
        a, b = np.array(res[c]).copy(), np.array(ref).copy()
        a["_pad"] = 0; b["_pad"] = 0
        assert a.tobytes() == b.tobytes(), (seed, c)
