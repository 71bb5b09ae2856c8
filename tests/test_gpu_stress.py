"""GPU suite: mode S (a V-validator commit on EVERY header — BASELINE configs #4/#5; circuits/next_header.rs:25-47 per header)
through ONE C-ABI call per step, `bsx_dev_verify_commits`, and its sharding across ranks (SURVEY §8e "in S mode commits shard
with their headers"): per-signature verdicts, every commit result and the 128-byte fold of every rank's slice against the
oracle; the all-gather itself over gloo with two ranks on the one GPU of the box."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean(r):
    r = np.array(r).copy()
    r["_pad"] = 0
    return r.tobytes()


@pytest.mark.parametrize("world,J,B,V,tamper", [(1, 2, 16, 12, False), (2, 4, 16, 33, True), (4, 8, 8, 7, True), (8, 32, 64, 100, False), (1, 32, 64, 100, True),
                                                 (8, 32, 64, 512, True),       # BASELINE config #5 literally
                                                 (2, 32, 32, 5, True), (1, 16, 64, 3, True), (1, 5, 64, 9, False)])   # folds of 2 / 4 sub-trees, a ragged one (320 commits)
def test_commit_shards_vs_oracle(world, J, B, V, tamper):
    """Every rank's CommitShard of an N-GPU mode-S run on ONE GPU: its slice's ok bits, commit results and fold equal the
    oracle's; the concatenation of the folds (= the all-gather's result) gives the range verdict, incl. the global index of a
    tampered commit and a nil/absent mix."""
    import torch
    from blobstreamx_amd.stress import CommitShard, range_verdict
    nh = J * B
    w = synth.Workload(5, 1, J, B, v=V, mode="S", nil_permille=100 if tamper else 0, absent_permille=50 if tamper else 0)
    vals = w.validators.reshape(nh, V).copy()
    bad = None
    if tamper:
        bad = nh // 2 + 3
        k = int(np.nonzero(vals[bad]["is_signed"])[0][0])
        vals[bad, k]["signature"][9] ^= 8
        if nh >= 2000:        # 2048 x 100 at world 1 is verified in two forms in one launch (k_ed25519_verify_keyed_mixed: commits
            bad2 = nh - 40    # [0, 1920) one lane per signature, the rest on four): one bad signature in the second part too
            k2 = int(np.nonzero(vals[bad2]["is_signed"])[0][-1])
            vals[bad2, k2]["signature"][40] ^= 1
    if nh * V > 100_000:      # a million signatures: the oracle on every host thread (same function, orc_verify_commit per commit)
        import os
        rres_all, rok_all = oracle.bench_verify_commits(vals, w.commit_hashes, len(os.sched_getaffinity(0)))
        ref = [(rres_all[c], rok_all[c]) for c in range(nh)]
    else:
        ref = [oracle.verify_commit(vals[c], w.commit_hashes[c].tobytes()) for c in range(nh)]
    folds = []
    for g in range(world):
        sh = CommitShard(nh, V, rank=g, world=world)
        sh.upload(vals, w.commit_hashes)
        sh.step()
        sh.step()                                   # warm tables, same answer
        ok, res, fold = sh.download()
        for c in range(sh.n):
            rres, rok = ref[sh.first + c]
            assert (ok[c] == rok).all(), (g, c)
            assert _clean(res[c]) == _clean(rres), (g, c)
        want = oracle.commit_fold(np.array([ref[sh.first + c][0] for c in range(sh.n)], T.COMMIT_RESULT), sh.first)
        assert fold.tobytes() == want.tobytes(), g
        assert sh.gather().tobytes() == fold.tobytes() if world == 1 else True
        folds.append(fold)
        del sh
        torch.cuda.empty_cache()
    v = range_verdict(np.array(folds, T.COMMIT_FOLD))
    # expected verdict from the oracle's results (with nil / absent votes a commit may also miss the 2/3 rule)
    good = [bool(r["two_thirds_ok"]) and not r["n_bad_signature"] and not r["n_bad_message"] and not r["power_overflow"] for r, _ in ref]
    assert v["commits"] == nh and v["ok"] == sum(good)
    assert v["first_failing"] == (good.index(False) if False in good else None)
    assert v["signatures_ok"] == sum(int(r["n_signed"]) - int(r["n_bad_signature"]) for r, _ in ref)
    if tamper:
        assert not good[bad] and ref[bad][0]["n_bad_signature"] == 1
    else:
        assert v["all_ok"] and v["signatures_ok"] == nh * V


def test_two_steps_in_flight_give_the_same_answers():
    """CommitShard(n_sets=2): step i on buffer set i mod 2 and its own stream, un-joined (bench.py's mode-S loop).  After an odd and
    after an even number of steps the verdicts, results, fold and COMMIT units of the last step equal the one-set shard's, and the
    fold gathered one step late is the fold of THAT step's set."""
    from blobstreamx_amd.stress import CommitShard
    J, B, V = 4, 16, 33
    nh = J * B
    w = synth.Workload(7, 1, J, B, v=V, mode="S", nil_permille=80, absent_permille=40)
    vals = w.validators.reshape(nh, V).copy()
    k = int(np.nonzero(vals[11]["is_signed"])[0][0])
    vals[11, k]["signature"][3] ^= 2
    one = CommitShard(nh, V, with_witness=True, tally_beside=False)      # the self-contained call: everything on the one stream
    one.upload(vals, w.commit_hashes)
    one.step()
    ok1, res1, fold1 = one.download()
    cw1 = one.compact_of(range(nh))
    two = CommitShard(nh, V, with_witness=True, n_sets=2)                 # BSX_COMMITS_TALLY_BESIDE (default): trees on the side stream
    two.upload(vals, w.commit_hashes)
    prev = None
    for i in range(5):
        k_set = two.step()
        assert k_set == i % 2
        if prev is not None:
            assert two.gather(prev).tobytes() == np.array([fold1]).tobytes()
        prev = k_set
        if i in (2, 3, 4):
            ok2, res2, fold2 = two.download()
            assert (ok2 == ok1).all() and _clean(res2) == _clean(res1) and fold2.tobytes() == fold1.tobytes(), i
            assert (two.compact_of(range(nh)) == cw1).all(), i
    rres, rok = oracle.verify_commit(vals[11], w.commit_hashes[11].tobytes())
    assert (ok1[11] == rok).all() and rres["n_bad_signature"] == 1


def test_bench_mode_s_two_ranks_share_one_gpu():
    """bench.py --mode S --gpus 2 end to end over gloo with both ranks on the one GPU: commit slices, one all-gather of folds per
    step, barriers, max-over-ranks timing, oracle checks on every rank's slice, ONE JSON line from rank 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BSX_DIST_BACKEND="gloo", BSX_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "S", "--jobs", "8", "--batch", "32",
           "--validators", "40", "--cpu-seconds", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 6144, out.stdout[-2000:]
    short = json.loads(lines[0])                        # the compact line (last on stdout); the full object is the DETAIL line before it
    assert short["n_gpus"] == 2 and short["value"] > 0 and short["roofline"]["frac"] <= 1.0 and short["cpu_baseline"]["value"] > 0
    assert short["legs"]["stress"]["all_ok"] is True
    from bench_legs.line import detail_of
    d = detail_of(out.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] <= 1.0 and d["cpu_baseline"]["value"] > 0
    s = d["stress"]
    assert s["signatures"] == 8 * 32 * 40 and s["signatures_this_rank"] == 4 * 32 * 40
    assert s["range_verdict"]["all_ok"] and s["range_verdict"]["commits"] == 256
    assert s["checked_against_oracle"]["gathered_folds"] == 2
