"""CPU suite: the multi-GPU path's partition + exchange logic over gloo (world_size 2 and 4).

Each rank computes the records of ITS slice of the map jobs (with the oracle standing in for the HIP kernels — the
kernels themselves are covered by the GPU suite), folds them locally, runs the product's one collective
(blobstreamx_amd.engine.gather_partials) and the top fold; the owner's result must equal the single-process
prove_data_commitment of the full range (circuits/builder.rs:299-395)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
import synth
from blobstreamx_amd import types as T
from blobstreamx_amd.engine import all_gather_records, gather_partials, job_slice


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, J, B, R, n_blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = synth.Workload(8, R * world, J, B, v=1, n_blocks=n_blocks)     # same seed on every rank
        jf, jc = job_slice(J, rank, world)
        RT = R * world
        partial = np.zeros(RT, T.SUBCHAIN)
        for r in range(RT):
            S = int(w.first_height[r])
            recs = []
            for j in range(jf, jf + jc):
                bs = S + j * B
                # this rank only looks at the headers of its own slice (+1): heights [S + jf*B, S + (jf+jc)*B]
                lo, hi = jf * B, (jf + jc) * B + 1
                rc, inp = oracle.data_commitment_inputs(w.headers[r, lo:hi], S + lo, int(w.latest[r]), bs, bs + B, B)
                assert rc == T.OK
                rc, rec, _ = oracle.prove_subchain(B, inp["start_header"], inp["end_header"], inp["data_hash_proofs"],
                                                   inp["last_block_id_proofs"], bs, bs + B, int(w.ranges[r]["end_block"]),
                                                   bytes(w.ranges[r]["end_header_hash"]))
                recs.append(rec)
            _, partial[r], _ = oracle.reduce(np.array(recs, T.SUBCHAIN))
        flat = torch.from_numpy(partial.view(np.uint8).reshape(-1).copy())
        top = gather_partials(flat, rank, world, R)
        # the branch bench.py's gloo path takes at N > 1 (engine.torch_allgather inside the bsx_pipeline_set_allgather callback): the collective issued with
        # async_op=True into a preallocated buffer, finished by work.wait() — must deliver the same [rank][range] image
        g_sync = all_gather_records(flat, world, RT)
        g_async, work = all_gather_records(flat, world, RT, out_gathered=torch.full((world * RT * 128,), 0xEE, dtype=torch.uint8), async_op=True)
        assert work is not None
        work.wait()
        assert torch.equal(g_sync, g_async)
        assert torch.equal(g_async.view(world, RT, 128)[rank], flat.view(RT, 128))
        top = top.numpy().view(T.SUBCHAIN).reshape(R, world)
        res = []
        for k in range(R):
            _, out, _ = oracle.reduce(top[k])
            res.append(bytes(out["data_merkle_root"]) + bytes(out["end_header"]) + int(out["end_block"]).to_bytes(8, "big") +
                       int(out["assert_fail"]).to_bytes(4, "big"))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,J,B,n_blocks", [(2, 4, 8, 32), (2, 4, 8, 19), (4, 8, 4, 32), (2, 2, 16, 5)])
def test_sharded_map_reduce_equals_single_process(world, J, B, n_blocks):
    R = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, J, B, R, n_blocks, q)) for rk in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = synth.Workload(8, R * world, J, B, v=1, n_blocks=n_blocks)
    for rk in range(world):
        for k in range(R):
            r = rk * R + k
            rc, ref = oracle.prove_data_commitment(J, B, w.ranges[r:r + 1], w.headers[r], int(w.first_height[r]), int(w.latest[r]))
            assert rc == T.OK
            want = ref["data_commitment"] + bytes(ref["result"]["end_header"]) + int(ref["result"]["end_block"]).to_bytes(8, "big") + \
                (0).to_bytes(4, "big")
            assert got[rk][k] == want, (rk, k)


def test_job_slice_rules():
    assert job_slice(32, 0, 8) == (0, 4) and job_slice(32, 7, 8) == (28, 4)
    assert job_slice(32, 0, 1) == (0, 32)
    with pytest.raises(AssertionError):
        job_slice(32, 0, 3)
    with pytest.raises(AssertionError):
        job_slice(24, 0, 4)     # 6 jobs per rank is not a power-of-two subtree


# ------------------------------------------------------------------------------------------------ mode S across ranks
def _worker_s(rank, world, port, nh, V, bad_commit, q):
    """Mode S sharded (BASELINE config #5): rank g verifies commits [g*nh/world, (g+1)*nh/world) — the oracle stands
    in for bsx_dev_verify_commits' kernels here, the product's slicing, fold layout and collective are the real ones."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blobstreamx_amd.stress import all_gather_folds, commit_slice, range_verdict
        w = synth.Workload(5, 1, 2, nh // 2, v=V, mode="S")
        vals = w.validators.reshape(nh, V).copy()
        if bad_commit is not None:
            vals[bad_commit, 1]["signature"][3] ^= 1
        first, n = commit_slice(nh, rank, world)
        res = np.zeros(n, T.COMMIT_RESULT)
        for c in range(n):
            res[c], _ = oracle.verify_commit(vals[first + c], w.commit_hashes[first + c].tobytes())
        fold = oracle.commit_fold(res, first)
        folds = all_gather_folds(torch.from_numpy(np.frombuffer(fold.tobytes(), np.uint8).copy()), world)
        folds = folds.numpy().reshape(-1).view(T.COMMIT_FOLD)
        q.put((rank, folds.tobytes(), range_verdict(folds)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_commit", [(2, None), (2, 11), (4, 5)])
def test_mode_s_commits_shard_with_their_headers(world, bad_commit):
    nh, V = 16, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_s, args=(rk, world, port, nh, V, bad_commit, q)) for rk in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: every slice's fold from the oracle's verify_commit of the whole range
    from blobstreamx_amd.stress import commit_slice
    w = synth.Workload(5, 1, 2, nh // 2, v=V, mode="S")
    vals = w.validators.reshape(nh, V).copy()
    if bad_commit is not None:
        vals[bad_commit, 1]["signature"][3] ^= 1
    res = np.zeros(nh, T.COMMIT_RESULT)
    for c in range(nh):
        res[c], _ = oracle.verify_commit(vals[c], w.commit_hashes[c].tobytes())
    want = b"".join(oracle.commit_fold(res[f:f + n], f).tobytes() for f, n in (commit_slice(nh, g, world) for g in range(world)))
    for rank, folds, verdict in got:
        assert folds == want, rank                           # every rank holds every slice's fold after the one collective
        assert verdict["commits"] == nh and verdict["all_ok"] == (bad_commit is None)
        assert verdict["first_failing"] == bad_commit
        assert verdict["ok"] == nh - (0 if bad_commit is None else 1)
    assert len({v["root_of_roots"] for _, _, v in got}) == 1


def test_commit_fold_is_sensitive_to_every_field_and_index():
    """The fold is a checksum of checksums: any bit of any commit result, the commit's position, or the slice offset moves it."""
    w = synth.Workload(5, 1, 2, 4, v=4, mode="S")
    res = np.zeros(8, T.COMMIT_RESULT)
    for c in range(8):
        res[c], _ = oracle.verify_commit(w.validators.reshape(8, 4)[c], w.commit_hashes[c].tobytes())
    res[1]["trusted_signed_power"] += 1          # the synthetic commits are signed by one set: make one result distinct for the swap below
    base = oracle.commit_fold(res, 0)
    assert base["n_commits"] == 8 and base["n_ok"] == 8 and base["first_failing"] == 0xffffffff and base["n_signatures_ok"] == 32
    seen = {bytes(base["root"])}
    for field in ("validators_hash", "total_power", "signed_power", "trusted_signed_power", "n_enabled", "n_signed", "n_bad_signature",
                  "first_bad_signature", "n_bad_message", "two_thirds_ok", "power_overflow"):
        r2 = res.copy()
        if field == "validators_hash":
            r2[3][field][7] ^= 1
        else:
            r2[3][field] ^= 1
        seen.add(bytes(oracle.commit_fold(r2, 0)["root"]))
    seen.add(bytes(oracle.commit_fold(res, 8)["root"]))                      # same results at another offset
    seen.add(bytes(oracle.commit_fold(res[[1, 0, 2, 3, 4, 5, 6, 7]], 0)["root"]))  # swapped
    seen.add(bytes(oracle.commit_fold(res[:7], 0)["root"]))                  # padded slice
    assert len(seen) == 15
    r2 = res.copy(); r2[5]["_pad"] = 7
    assert bytes(oracle.commit_fold(r2, 0)["root"]) == bytes(base["root"])    # padding bytes are not part of the record


def _run_bench(argv, extra_env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BSX_DIST_BACKEND="gloo", **(extra_env or {}))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, [json.loads(ln) for ln in lines]


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the environment must not run as N = 1 — it re-executes itself
    under torch.distributed.run with 2 ranks (the CPU-only dry path: process group + one all-gather, no GPU work) and reports the
    world that really ran; modes F and S share the launcher."""
    for argv in (["--gpus", "2", "--dry-run"], ["--gpus", "2", "--mode", "S", "--dry-run"]):
        out, lines = _run_bench(argv)
        assert out.returncode == 0, out.stderr[-2000:]
        assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks"] == [0, 1] and lines[0]["processes"] == 2, lines
        assert lines[0]["launched_by"] == "bench.py itself"
    out, lines = _run_bench(["--gpus", "1", "--dry-run"])
    assert out.returncode == 0 and lines[0]["n_gpus"] == 1 and lines[0]["launched_by"] == "the caller / an external launcher"
