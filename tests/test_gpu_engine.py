"""GPU suite: the device-resident batched pipeline bench.py times — `bsx_pipeline_*` of the C ABI (csrc/pipeline.hip) through
its thin ctypes wrapper blobstreamx_amd/engine.py — against the oracle: public outputs, per-job records, commit results and
the complete Goldilocks witness of every range of a batch."""
import numpy as np
import pytest

import oracle
import synth
from blobstreamx_amd import types as T

pytestmark = pytest.mark.gpu


def _rec(r):
    r = np.array(r, dtype=T.SUBCHAIN).copy()
    r["_pad"] = 0
    return r.tobytes()


# BASELINE configs: #2 64 = 2x32 / V=100, #3 1024 = 32x32, #4 2048 = 32x64, #5 2048 / V=512 (here on one GPU; the sharded
# form of #4/#5 is test_sharded_engines_on_one_gpu)
@pytest.mark.parametrize("J,B,V,R,n_blocks", [(2, 32, 100, 3, 64), (32, 64, 100, 2, 2048), (8, 32, 20, 4, 131), (32, 32, 100, 2, 1024),
                                              (32, 64, 512, 1, 2048)])
def test_engine_batch_vs_oracle(J, B, V, R, n_blocks):
    from blobstreamx_amd.engine import HeaderRangeEngine
    w = synth.Workload(4, R, J, B, v=V, n_blocks=n_blocks)
    eng = HeaderRangeEngine(J, B, V, R)
    eng.upload_workload(w)
    eng.step()
    eng.step()      # a second pass over the same buffers must give the same answer (status words re-zeroed, no stale state)
    res = eng.download()
    wm, wrl, _ = eng.witness_numpy()
    ml, rl = T.map_layout(B), T.reduce_layout()
    nm, nr = J * int(ml["n_elements"]), (J - 1) * int(rl["n_elements"])
    assert res["header_status"] == 0 and res["assemble_status"] == 0
    for r in range(R):
        rc, out, cres, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                                w.validators[r], w.trusted[r], want_witness=True)
        assert rc == T.OK
        assert res["output64"][r].tobytes() == out, r
        assert res["range_status"][r] == 0 and res["skip_status"][r] == 0
        got = np.array(res["commit"][r]).copy(); got["_pad"] = 0
        want = np.array(cres).copy(); want["_pad"] = 0
        assert got.tobytes() == want.tobytes()
        ref = oracle.expand_range_witness(J, B, cw)
        assert (wm[r * nm:(r + 1) * nm] == ref[:nm]).all(), r
        assert (wrl[r * nr:(r + 1) * nr] == ref[nm:]).all(), r


def _check_pipelined_against_oracle(pe, w, J, B):
    """Every range of a PipelinedEngines batch vs the oracle: public output, statuses, commit result, per-job records and
    the complete map + reduce Goldilocks witness (the same assertions as test_engine_batch_vs_oracle, plus records and
    the failure verdicts of test_engine_reports_failures_per_range)."""
    res = pe.download()
    ml, rl = T.map_layout(B), T.reduce_layout()
    nm, nr = J * int(ml["n_elements"]), (J - 1) * int(rl["n_elements"])
    assert res["header_status"] == 0 and res["assemble_status"] == 0
    wits = [pe.witness_numpy(e) for e in range(pe.E)]
    for r in range(pe.R):
        e, k = divmod(r, pe.Rc)
        rc, out, cres, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                                w.validators[r], w.trusted[r], want_witness=True)
        mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
        assert mine == rc, (r, mine, rc)
        assert res["output64"][r].tobytes() == out, r
        got = np.array(res["commit"][r]).copy(); got["_pad"] = 0
        want = np.array(cres).copy(); want["_pad"] = 0
        assert got.tobytes() == want.tobytes(), r
        ctx = w.ranges[r:r + 1].copy()
        ctx["end_header_hash"][0] = np.frombuffer(out[:32], np.uint8)
        _, ref = oracle.prove_data_commitment(J, B, ctx, w.headers[r], int(w.first_height[r]), int(w.latest[r]))
        assert [_rec(x) for x in res["records"][r]] == [_rec(x) for x in ref["records"]], r
        full = oracle.expand_range_witness(J, B, cw)
        wm, wrl, _ = wits[e]
        assert (wm[k * nm:(k + 1) * nm] == full[:nm]).all(), r
        assert (wrl[k * nr:(k + 1) * nr] == full[nm:]).all(), r
    return res


@pytest.mark.parametrize("J,B,V,R,n_blocks", [(32, 64, 100, 4, 2048), (32, 32, 100, 4, 1024), (8, 32, 20, 6, 131)])
def test_pipelined_engines_multi_step_vs_oracle(J, B, V, R, n_blocks):
    """THE object bench.py times: PipelinedEngines with two chunks on separate streams, event tokens, the commit check
    deferred onto side streams and double-buffered by pass parity.  Three consecutive step()s WITHOUT a join in between
    (cross-step races — step i+1's hint / header hashing vs step i's side-stream commit check or expansion — would
    corrupt a status, a record or the witness), then every range is diffed against the oracle.  One tampered range per
    chunk: the per-range verdicts must be the oracle's and must not leak into the neighbours."""
    from blobstreamx_amd.engine import PipelinedEngines
    w = synth.Workload(4, R, J, B, v=V, n_blocks=n_blocks)
    w.headers[1, 9]["hash"][1][5] ^= 1                     # chunk 0: range 1's chain breaks at header 9
    w.validators[R - 1, 3]["signature"][0] ^= 2            # chunk 1: the last range carries one bad signature
    pe = PipelinedEngines(J, B, V, R, n_engines=2)
    pe.upload_workload(w)
    for _ in range(3):
        pe.step()
    res = _check_pipelined_against_oracle(pe, w, J, B)
    assert res["range_status"][1] != 0 and res["skip_status"][R - 1] == T.ERR_BAD_SIGNATURE
    assert not res["range_status"][[r for r in range(R) if r != 1]].any()
    # new inputs into the same buffers, again without draining the pipeline first: nothing of the old batch may survive
    w2 = synth.Workload(5, R, J, B, v=V, n_blocks=n_blocks)
    pe.step()
    pe.upload_workload(w2)          # upload synchronises the device
    for _ in range(2):
        pe.step()
    res = _check_pipelined_against_oracle(pe, w2, J, B)
    assert not res["range_status"].any() and not res["skip_status"].any()


@pytest.mark.parametrize("J,B,V,R,p,n_engines,witness", [(8, 32, 20, 12, 100, 2, True), (4, 16, 9, 16, 1000, 1, False), (8, 32, 100, 6, 10, 2, True)])
def test_validator_sets_that_change_between_the_ranges_of_a_chunk_vs_oracle(J, B, V, R, p, n_engines, witness):
    """VERDICT r4 missing #5: validator sets differ between the ranges of one chunk — that is what `skip` exists for
    (/root/reference/circuits/header_range.rs:42-48, /root/reference/circuits/fetcher.rs:60-87).  synth rotates p / 1000 of the slots
    from range to range (p = 1000: every range its own set); the chunk's fixed-key table holds one row per DISTINCT public key
    (csrc/keycache.h), so no signature falls back to the generic kernel — and whatever the map, every range's output, statuses,
    commit result, records and witness are the oracle's.  One forged signature by a ROTATED key and one range signed by a set the
    headers do not commit to keep the failure paths honest; a second upload re-uses the table with further keys."""
    from blobstreamx_amd.engine import AlternatingPipelines, PipelinedEngines
    w = synth.Workload(21, R, J, B, v=V, rotate_permille=p)
    changed = [int((w.validators[r]["pubkey"] != w.validators[0]["pubkey"]).any(axis=1).sum()) for r in range(R)]
    assert changed[-1] >= 1 and changed[0] == 0
    r_bad = R - 1
    slot = int(np.nonzero((w.validators[r_bad]["pubkey"] != w.validators[0]["pubkey"]).any(axis=1))[0][0])
    w.validators[r_bad, slot]["signature"][9] ^= 8                  # a rotated key's signature, forged
    w.validators[2] = w.validators[1]; w.validators[2]["message"] = w.validators[1]["message"]   # range 2 carries range 1's commit: wrong block hash
    pe = PipelinedEngines(J, B, V, R, n_engines=n_engines, with_witness=True) if witness else AlternatingPipelines(2, J, B, V, R, n_engines=n_engines, with_witness=False)
    pe.upload_workload(w)
    for _ in range(2):
        pe.step()
    if witness:
        res = _check_pipelined_against_oracle(pe, w, J, B)
    else:
        res = pe.download()
        for r in range(R):
            rc, out, cres, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])
            mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
            assert mine == rc and res["output64"][r].tobytes() == out, (r, mine, rc)
            got = np.array(res["commit"][r]).copy(); got["_pad"] = 0
            want = np.array(cres).copy(); want["_pad"] = 0
            assert got.tobytes() == want.tobytes(), r
    assert res["skip_status"][r_bad] == T.ERR_BAD_SIGNATURE and res["skip_status"][2] != 0
    assert not res["skip_status"][[r for r in range(R) if r not in (2, r_bad)]].any()
    # the next upload: the chain has moved on (other rotation seed), most keys are already in the table
    w2 = synth.Workload(22, R, J, B, v=V, rotate_permille=p)
    pe.upload_workload(w2)
    pe.step()
    res = pe.download()
    for r in range(R):
        rc, out, cres, _ = oracle.header_range(J, B, w2.input48(r), w2.headers[r], int(w2.first_height[r]), int(w2.latest[r]), w2.validators[r], w2.trusted[r])
        assert rc == T.OK and res["skip_status"][r] == 0 and res["range_status"][r] == 0 and res["output64"][r].tobytes() == out, r


@pytest.mark.parametrize("J,B,R,n_blocks,top", [(32, 64, 32, 2048 - 37, 8), (32, 32, 32, 1024, 4), (64, 16, 16, 1024 - 5, 2), (8, 128, 128, 1024 - 77, 16),
                                                  (4, 256, 256, 1024 - 300, 32)])
def test_commitment_tree_tops_in_their_own_launch_vs_oracle(J, B, R, n_blocks, top):
    """A chunk of >= 1024 map jobs: k_batch_finish stops at the level that no longer fills half a wave of a workgroup and
    k_batch_top<top> (one lane per job) hashes the rest and writes the batch tail.  Outputs, statuses and records of every range,
    the complete map + reduce witness of five of them (a tampered one, a range ending inside a batch: masked tree nodes) vs the
    oracle."""
    from blobstreamx_amd.engine import PipelinedEngines, BUF_WITNESS_MAP, BUF_WITNESS_REDUCE_LOCAL
    assert R * J >= 1024
    V = 6
    w = synth.Workload(11, R, J, B, v=V, n_blocks=n_blocks)
    w.headers[3, n_blocks // 2]["hash"][6][3] ^= 1          # data_hash of one header of range 3
    pe = PipelinedEngines(J, B, V, R, n_engines=1)
    pe.upload_workload(w)
    pe.step()
    pe.step()
    res = pe.download()
    assert res["header_status"] == 0 and res["assemble_status"] == 0
    ml, rl = T.map_layout(B), T.reduce_layout()
    nm, nr = J * int(ml["n_elements"]), (J - 1) * int(rl["n_elements"])
    wm, wr = pe.buffer(0, BUF_WITNESS_MAP, i64=True), pe.buffer(0, BUF_WITNESS_REDUCE_LOCAL, i64=True)
    for r in range(R):
        full = r in (0, 3, 7, R // 2, R - 1)
        rc, out, cres, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                                w.validators[r], w.trusted[r], want_witness=full)
        mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
        assert mine == rc, (r, mine, rc)
        assert res["output64"][r].tobytes() == out, r
        ctx = w.ranges[r:r + 1].copy()
        ctx["end_header_hash"][0] = np.frombuffer(out[:32], np.uint8)
        _, ref = oracle.prove_data_commitment(J, B, ctx, w.headers[r], int(w.first_height[r]), int(w.latest[r]))
        assert [_rec(x) for x in res["records"][r]] == [_rec(x) for x in ref["records"]], r
        if full:
            want = oracle.expand_range_witness(J, B, cw)
            assert (wm[r * nm:(r + 1) * nm].cpu().numpy().view(np.uint64) == want[:nm]).all(), r
            assert (wr[r * nr:(r + 1) * nr].cpu().numpy().view(np.uint64) == want[nm:]).all(), r
    assert res["range_status"][3] != 0 and not np.delete(res["range_status"], 3).any()


@pytest.mark.parametrize("n_engines", [1, 2])
def test_alternating_pipelines_compact_only_vs_oracle(n_engines):
    """The compact-only leg's object: ONE bsx_pipeline with two buffer sets (bsx_pipeline_config.n_sets = 2; step i on set i mod 2)
    stepped WITHOUT joins (step i + 1 starts while step i's kernels drain; a token serialises the header hashings), one-launch
    prove_subchain, no witness.  Public outputs, statuses, commit results and per-job records of every range against the oracle,
    after an odd and after an even number of steps (= from both sets), with one tampered range."""
    from blobstreamx_amd.engine import AlternatingPipelines
    J, B, V, R, n_blocks = 32, 64, 100, 4, 2048
    w = synth.Workload(4, R, J, B, v=V, n_blocks=n_blocks)
    w.headers[2, 700]["hash"][1][5] ^= 1
    ap = AlternatingPipelines(2, J, B, V, R, n_engines=n_engines, with_witness=False)
    ap.upload_workload(w)
    refs = []
    for r in range(R):
        rc, out, cres, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])
        ctx = w.ranges[r:r + 1].copy()
        ctx["end_header_hash"][0] = np.frombuffer(out[:32], np.uint8)
        _, ref = oracle.prove_data_commitment(J, B, ctx, w.headers[r], int(w.first_height[r]), int(w.latest[r]))
        refs.append((rc, out, cres, ref))
    outs = []
    for n_steps in (5, 2):                                # steps 0..4: the last ran on set 0; one more below, then steps 6, 7: set 1
        for _ in range(n_steps):
            ap.step()
        res = ap.download()
        outs.append(res)
        assert res["header_status"] == 0 and res["assemble_status"] == 0
        for r in range(R):
            rc, out, cres, ref = refs[r]
            mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
            assert mine == rc, (r, mine, rc)
            assert res["output64"][r].tobytes() == out, r
            got = np.array(res["commit"][r]).copy(); got["_pad"] = 0
            want = np.array(cres).copy(); want["_pad"] = 0
            assert got.tobytes() == want.tobytes(), r
            assert [_rec(x) for x in res["records"][r]] == [_rec(x) for x in ref["records"]], r
        assert res["range_status"][2] != 0 and not res["range_status"][[0, 1, 3]].any()
        ap.step()                                         # one more: the next download reads the OTHER set
    assert outs[0]["output64"].tobytes() == outs[1]["output64"].tobytes()


@pytest.mark.parametrize("ed_path,commit_with", [("generic", "expand"), ("keyed", "expand"), ("keyed", "hash"), ("generic", "hash")])
def test_engine_reports_failures_per_range(ed_path, commit_with):
    """Both forms of the signature check (per-signature / per-validator tables) and both placements of the commit
    side stream must report the same per-range verdicts as the oracle."""
    from blobstreamx_amd.engine import HeaderRangeEngine
    J, B, V, R = 4, 8, 12, 4
    w = synth.Workload(6, R, J, B, v=V)
    w.headers[1, 9]["hash"][1][5] ^= 1            # range 1: chain breaks at header 9
    w.validators[2, 3]["signature"][0] ^= 2       # range 2: one bad signature
    w.trusted[3, 0]["voting_power"] += 7          # range 3: trusted set no longer matches the header
    eng = HeaderRangeEngine(J, B, V, R, ed_path=ed_path, commit_with=commit_with)
    eng.upload_workload(w)
    eng.step()
    res = eng.download()
    assert res["range_status"][0] == 0 and res["skip_status"][0] == T.OK
    assert res["range_status"][1] != 0 and res["skip_status"][1] == T.OK
    assert res["skip_status"][2] == T.ERR_BAD_SIGNATURE and res["commit"][2]["first_bad_signature"] == 3
    assert res["skip_status"][3] == T.ERR_ASSERT
    for r in range(R):
        rc = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])[0]
        mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
        assert mine == rc, (r, mine, rc)


@pytest.mark.parametrize("world,J,B,R,n_blocks,V,E", [(2, 4, 8, 2, 32, 12, 1), (4, 8, 8, 2, 37, 12, 2), (8, 32, 64, 1, 2048, 12, 1), (2, 2, 32, 3, 64, 12, 1),
                                                      (2, 32, 64, 2, 2048, 100, 2), (8, 32, 64, 1, 2048, 512, 1)])
def test_sharded_engines_on_one_gpu(world, J, B, R, n_blocks, V, E):
    """Every rank's pipeline of an N-GPU run, executed on ONE GPU with the all-gather delivered by a callback that hands every
    rank the concatenation of all ranks' partial buffers (what ncclAllGather returns): job slices + header_first_rel + local
    fold + exchange stream + top fold + owner's commit/finalize/witness must reproduce the oracle for every range — public
    output, statuses, commit result, this rank's records and map-job witness of EVERY range, and the owner's top reduce nodes.
    Includes the BASELINE shapes: config #4 (32x64, V = 100, two pipelined chunks) and #5 (V = 512, world 8).  The real
    collective is covered over gloo in tests/test_distributed_cpu.py and test_bench_two_ranks_share_one_gpu."""
    from blobstreamx_amd.engine import PipelinedEngines, run_world_on_one_gpu
    w = synth.Workload(9, R * world, J, B, v=V, n_blocks=n_blocks)
    engs = [PipelinedEngines(J, B, V, R, n_engines=E, rank=g, world=world) for g in range(world)]
    for e in engs:
        e.upload_workload(w)
    run_world_on_one_gpu(engs)
    ml, rl = T.map_layout(B), T.reduce_layout()
    jc = J // world
    nm = jc * int(ml["n_elements"])
    nt = (world - 1) * int(rl["n_elements"])
    refs = [oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r],
                                want_witness=True) for r in range(R * world)]
    fulls = [oracle.expand_range_witness(J, B, ref[3]) for ref in refs]
    nmap = J * int(ml["n_elements"])
    for g, e in enumerate(engs):
        out = e.download()
        assert out["header_status"] == 0 and out["assemble_status"] == 0
        for k in range(R):
            rc, ref_out, cres, _ = refs[g * R + k]
            assert rc == T.OK
            assert out["output64"][k].tobytes() == ref_out, (g, k)
            assert out["range_status"][k] == 0 and out["skip_status"][k] == 0, (g, k, out["range_status"], out["skip_status"], out["commit"][k])
            got = np.array(out["commit"][k]).copy(); got["_pad"] = 0
            want = np.array(cres).copy(); want["_pad"] = 0
            assert got.tobytes() == want.tobytes(), (g, k)
        for c in range(E):
            wm, _, wrt = e.witness_numpy(c)
            for i, r in enumerate(e.sel(c)):
                # map-job witnesses of this rank's slice of EVERY range equal the oracle's jobs [g*jc, (g+1)*jc)
                assert (wm[i * nm:(i + 1) * nm] == fulls[r][g * nm:(g + 1) * nm]).all(), (g, r)
                ctx = w.ranges[r:r + 1].copy()
                ctx["end_header_hash"][0] = np.frombuffer(refs[r][1][:32], np.uint8)
                _, ref = oracle.prove_data_commitment(J, B, ctx, w.headers[r], int(w.first_height[r]), int(w.latest[r]))
                assert [_rec(x) for x in out["records"][r]] == [_rec(x) for x in ref["records"][g * jc:(g + 1) * jc]], (g, r)
            # the owner's top reduce nodes (the last log2(world) levels of the reference's tree) = the tail of the oracle's
            # reduce section: level order, so the top world-1 nodes are its last world-1 entries
            for k in range(e.Rc):
                r = g * R + c * e.Rc + k
                assert (wrt[k * nt:(k + 1) * nt] == fulls[r][nmap + (J - 1 - (world - 1)) * int(rl["n_elements"]):]).all(), (g, r)


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 code path end to end — RANK/WORLD_SIZE plumbing, job slices, one all-gather per pipelined chunk,
    strided top fold, barriers, max-over-ranks timing, the correctness gate on every owned range — with two ranks on the
    ONE GPU of the test box over gloo (the driver's multi-GPU runs use one rank per GPU over RCCL; only the transport
    differs, see engine.all_gather_records)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BSX_DIST_BACKEND="gloo", BSX_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--ranges", "8", "--no-cpu-baseline", "--no-stress"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE compact JSON line ...
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 6144      # ... the LAST line of stdout, < 6 KB
    short = json.loads(lines[0])
    from bench_legs.line import detail_of
    d = detail_of(out.stdout)                           # the full object: the `DETAIL {...}` line before it
    assert short["n_gpus"] == 2 and short["value"] == d["value"] and short["config"]["multi_gpu"]["per_rank_ms_per_step"]
    assert short["roofline"]["frac"] > 0 and short["config"]["nccl_ranks"] == 2
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["ranges_per_gpu"] == 8 and d["config"]["headers_per_step"] == 2 * 8 * 2048
    # bsx_pipeline_autotune ran as a collective: both ranks agreed on the steps per trial through the all-gather callback
    assert d["config"]["stream_autotune"]["n_trials"] == 26 and d["config"]["stream_autotune"]["steps_per_trial"] >= 3
    # what makes the first hardware scaling run diagnosable: the world that ran, which collective, every rank's own step
    # time, and the all-gather timed with HIP events on the library's exchange stream — one per chunk per timed step on every rank
    mg = d["config"]["multi_gpu"]
    assert d["config"]["nccl_ranks"] == d["n_gpus"] == 2 and d["config"]["collective"]
    assert len(mg["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in mg["per_rank_ms_per_step"])
    assert mg["allgathers_timed_per_rank"] == [3 * d["config"]["pipelined_chunks"]] * 2
    for k in ("avg", "min", "median", "max"):
        assert len(mg["allgather_us_per_chunk"][k]) == 2 and all(v > 0 for v in mg["allgather_us_per_chunk"][k]), (k, mg)
    assert all(lo <= md <= hi for lo, md, hi in zip(mg["allgather_us_per_chunk"]["min"], mg["allgather_us_per_chunk"]["median"], mg["allgather_us_per_chunk"]["max"]))
    assert d["ms_per_step"] >= max(mg["per_rank_ms_per_step"]) * 0.999


def test_expansion_kernel_variants_agree():
    """k_expand_witness has ten instantiations in the EXPERIMENTS build (staging chunk 256/512/1024/2048/4096 x plain /
    non-temporal stores, chosen by BSX_EXPAND_CHUNK / BSX_EXPAND_NT when libbsx_exp.so is loaded; the product library holds only
    the one it launches: 256, non-temporal).  Every one of them, and a capped grid, must emit the oracle's witness bit for bit; each
    runs in its own process because the choice is read once — and reports bsx_debug_last_launch_form(1), which must be the
    requested instantiation (a run that silently takes the default fails)."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    J, B, V, R = 8, 32, 10, 3
    w = synth.Workload(9, R, J, B, v=V, n_blocks=200)
    ml, rl = T.map_layout(B), T.reduce_layout()
    want = hashlib.sha256()
    for r in range(R):
        rc, _, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]),
                                           w.validators[r], w.trusted[r], want_witness=True)
        assert rc == T.OK
        want.update(oracle.expand_range_witness(J, B, cw)[:J * int(ml["n_elements"])].tobytes())
    snippet = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "import synth\n"
        "from blobstreamx_amd.engine import HeaderRangeEngine\n"
        "w = synth.Workload(9, %d, %d, %d, v=%d, n_blocks=200)\n"
        "e = HeaderRangeEngine(%d, %d, %d, %d); e.upload_workload(w); e.step()\n"
        "m, _, _ = e.witness_numpy(); print('WITNESS', hashlib.sha256(m.tobytes()).hexdigest())\n"
        "import ctypes; from blobstreamx_amd import _lib; L = _lib.lib(); L.bsx_debug_last_launch_form.restype = ctypes.c_uint32\n"
        "print('FORM', hex(L.bsx_debug_last_launch_form(ctypes.c_uint32(1))))\n" % (root, R, J, B, V, J, B, V, R))
    exp_lib = os.path.join(root, "blobstreamx_amd", "lib", "libbsx_exp.so")
    assert os.path.exists(exp_lib), "libbsx_exp.so is not built: python -c 'import __graft_entry__ as g; g.build()'"
    variants = [(c, nt, "") for c in (256, 512, 1024, 2048, 4096) for nt in (0, 1)] + [(256, 1, "7"), (2048, 0, "1")]
    for chunk, nt, cap in variants:
        env = dict(os.environ, BSX_EXPAND_CHUNK=str(chunk), BSX_EXPAND_NT=str(nt), BSX_LIB_OVERRIDE=exp_lib)
        env.pop("BSX_EXPAND_BLOCKS", None)
        if cap:
            env["BSX_EXPAND_BLOCKS"] = cap
        out = subprocess.run([sys.executable, "-c", snippet], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (chunk, nt, cap, out.stderr[-2000:])
        got = [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("WITNESS")]
        assert got == [want.hexdigest()], (chunk, nt, cap)
        # the last expansion the engine launched (the reduce nodes', same instantiation) ran the REQUESTED form; the capped grid
        # bit is set for the map-job launch only, so it is not part of this check
        form = [int(ln.split()[1], 16) for ln in out.stdout.splitlines() if ln.startswith("FORM")]
        assert form and (form[0] & 0x1ffff) == (chunk | (nt << 16)), (chunk, nt, cap, form)


def test_pipeline_driven_from_cpp_without_python(tmp_path):
    """the batched pipeline is behind the C ABI.  tests/hostcheck/pipeline_driver.cpp — a plain C++ program that
    includes include/bsx.h only and links libbsx.so — creates a 2-chunk pipeline with witness + commit check, uploads, enqueues
    three steps WITHOUT joins and fetches the results; they must be the oracle's for every range (one tampered range per chunk)."""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "hostcheck"), "pipeline_driver"], check=True, capture_output=True)
    J, B, V, R, E, steps = 8, 32, 20, 4, 2, 3
    w = synth.Workload(41, R, J, B, v=V)
    w.headers[1, 9]["hash"][1][5] ^= 1
    w.validators[R - 1, 3]["signature"][0] ^= 2
    outs, rs, ss = [], [], []
    for r in range(R):
        rc, out, _, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])
        outs.append(out)
        ss.append(rc if rc in (T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER, T.ERR_BAD_ARG) else 0)
        rs.append(1 if rc == T.ERR_ASSERT else 0)
    assert rs == [0, 1, 0, 0] and ss[R - 1] == T.ERR_BAD_SIGNATURE
    cid = b"celestia"
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(struct.pack("<10I56s", 0x42535850, J, B, V, R, E, w.hpr, steps, 1 | 2, len(cid), cid))
        for a in (w.headers[:R], w.ranges[:R], np.ascontiguousarray(w.latest[:R], np.uint64), w.validators[:R], w.trusted[:R]):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(b"".join(outs))
        f.write(np.array(rs, np.uint32).tobytes())
        f.write(np.array(ss, np.uint32).tobytes())
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([os.path.join(root, "tests", "hostcheck", "pipeline_driver"), str(case)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "PIPELINE_DRIVER_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])
    assert f"ranges={R} steps={steps} chunks={E}" in out.stdout
    # bsx_prepare_process (called by the driver before its first HIP call) asked for 16 hardware queues; what the pool's streams
    # really got is MEASURED by the library (groups of streams whose kernels run one after the other) and reported by
    # bsx_pipeline_autotune: more than HIP's default of 4 (queues that share a dispatch pipe still serialise, so it
    # need not be 16)
    import re
    hq = int(re.search(r"hw_queues=(\d+)", out.stdout).group(1))
    assert 4 < hq <= 16, out.stdout
    # RCCL called from the C tier — a one-rank communicator made through bsx_rccl_*, ncclAllGather checked end to end
    # by bsx_pipeline_check_allgather before and after the steps; no Python, no torch in that process
    out = subprocess.run([os.path.join(root, "tests", "hostcheck", "pipeline_driver"), str(case), "--rccl"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "PIPELINE_DRIVER_OK" in out.stdout and "rccl=ok" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])


@pytest.mark.parametrize("J,B,V,R,leaf_len,cap_h", [(8, 32, 20, 4, 135, 4), (32, 64, 100, 2, 135, 4), (4, 16, 6, 6, 80, 2)])
def test_pipeline_caps_mode_multi_step_vs_oracle(J, B, V, R, leaf_len, cap_h):
    """BSX_PIPE_CAPS: the pipeline commits to every map-job witness with a Poseidon Merkle cap hashed STRAIGHT
    FROM THE COMPACT BYTES inside the step (no 64x image; here together with the materialised witness so that both can be
    compared).  Three un-joined steps over two chunks, then every cap node of every job against the oracle's own witness
    hashed by the oracle's own Poseidon, and the usual outputs / statuses."""
    from blobstreamx_amd.engine import PipelinedEngines
    w = synth.Workload(44, R, J, B, v=V)
    w.headers[1, 5]["hash"][1][9] ^= 4
    pe = PipelinedEngines(J, B, V, R, n_engines=2, with_witness=True, with_caps=True, leaf_len=leaf_len, cap_height=cap_h)
    pe.upload_workload(w)
    for _ in range(3):
        pe.step()
    res = pe.download()
    ml = T.map_layout(B)
    nel = int(ml["n_elements"])
    for e in range(pe.E):
        trees, caps = pe.caps_numpy(e)
        wm, _, _ = pe.witness_numpy(e)
        n_leaves = (trees.shape[1] + caps.shape[1]) // 2
        for i, r in enumerate(pe.sel(e)):
            rc, out, _, cw = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r],
                                                 want_witness=True)
            assert res["output64"][r].tobytes() == out and (res["range_status"][r] != 0) == (rc == T.ERR_ASSERT)
            full = oracle.expand_range_witness(J, B, cw)
            assert (wm[i * J * nel:(i + 1) * J * nel] == full[:J * nel]).all()
            for j in range(J):
                tree, cap = oracle.poseidon_merkle_tree(full[j * nel:(j + 1) * nel], leaf_len, n_leaves, min(cap_h, n_leaves.bit_length() - 1))
                assert (trees[i * J + j] == tree).all(), (r, j)
                assert (caps[i * J + j] == cap).all(), (r, j)
    assert res["range_status"][1] != 0 and not res["range_status"][[r for r in range(R) if r != 1]].any()


def test_rccl_all_gather_on_an_external_stream_world_1():
    """The collective the N > 1 pipeline hands to torch.distributed runs over RCCL on a stream the LIBRARY owns (wrapped as a
    torch ExternalStream).  One GPU cannot host two RCCL ranks, but a world of ONE exercises the same calls: process group
    over "nccl" (= RCCL), all_gather_into_tensor enqueued behind work on the external stream, results visible to work
    enqueued after it — both for the pipeline's record exchange and for mode S's folds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from blobstreamx_amd import _lib\n"
        "from blobstreamx_amd.engine import torch_allgather\n"
        "from blobstreamx_amd.stress import all_gather_folds\n"
        "_lib.lib(); torch.cuda.set_device(0); dev = torch.device('cuda:0')\n"
        "os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ.setdefault('MASTER_PORT', '29517')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "s = torch.cuda.Stream(device=dev)                      # stands in for the pipeline's exchange stream\n"
        "send = torch.zeros(4096 * 128, dtype=torch.uint8, device=dev); recv = torch.full_like(send, 0xEE)\n"
        "with torch.cuda.stream(s):\n"
        "    torch.cuda._sleep(20_000_000)                      # the collective must wait for this stream's earlier work\n"
        "    send.copy_((torch.arange(send.numel(), device=dev) % 251).to(torch.uint8))\n"
        "torch_allgather(dev, 1)(send, recv, s.cuda_stream)\n"
        "with torch.cuda.stream(s):\n"
        "    after = recv.clone()                               # enqueued after the callback returned: must see the gathered data\n"
        "s.synchronize()\n"
        "assert torch.equal(after, send) and int(after[250]) == 250\n"
        "# the pipeline's own buffers (library allocations, not torch's): the tensors the callback really sees at N > 1\n"
        "from blobstreamx_amd import engine as E\n"
        "pe = E.HeaderRangeEngine(2, 4, 4, 2, with_witness=False)\n"
        "ps, pr = pe.buffer(0, E.BUF_PARTIAL), pe.buffer(0, E.BUF_GATHERED)\n"
        "ps.copy_((torch.arange(ps.numel(), device=dev) % 199).to(torch.uint8)); pr.zero_(); torch.cuda.synchronize()\n"
        "torch_allgather(dev, 1)(ps, pr, s.cuda_stream); s.synchronize()\n"
        "assert torch.equal(ps, pr)\n"
        "f = all_gather_folds(send[:128], 1)\n"
        "assert f.shape == (1, 128) and torch.equal(f[0], send[:128])\n"
        "print('RCCL_WORLD1_OK', dist.get_backend())\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_WORLD1_OK nccl" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_rccl_from_the_c_tier_beside_torchs_own_process_group():
    """bench.py's N > 1 branch, at a world of ONE (one GPU cannot host two RCCL ranks): a torch.distributed process group over "nccl"
    is alive, a SECOND communicator is made through the C ABI (engine.c_rccl_comm -> bsx_rccl_get_unique_id / _comm_init_rank), the
    pipeline is told to call ncclAllGather itself (bsx_pipeline_set_rccl + bsx_pipeline_check_allgather), and steps still equal the
    oracle — RCCL bound at run time next to PyTorch's own copy."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from blobstreamx_amd import _lib, types as T\n"
        "from blobstreamx_amd import engine as E\n"
        "import synth, oracle\n"
        "_lib.lib(); torch.cuda.set_device(0); dev = torch.device('cuda:0')\n"
        "os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ.setdefault('MASTER_PORT', '29519')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "t = torch.ones(4, device=dev); dist.all_reduce(t)              # torch's own communicator is live\n"
        "J, B, V, R = 4, 8, 10, 4\n"
        "w = synth.Workload(3, R, J, B, v=V)\n"
        "pe = E.PipelinedEngines(J, B, V, R, n_engines=2)\n"
        "comm = E.c_rccl_comm(pe.ctx, 0, 1)\n"
        "pe.set_rccl(comm)                                               # includes bsx_pipeline_check_allgather\n"
        "pe.upload_workload(w); pe.step(); pe.step(); pe.check_allgather()\n"
        "res = pe.download()\n"
        "for r in range(R):\n"
        "    rc, out, _, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])\n"
        "    assert rc == 0 and res['output64'][r].tobytes() == out and res['skip_status'][r] == 0\n"
        "pe.close()\n"
        "assert _lib.lib().bsx_rccl_comm_destroy(comm) == 0\n"
        "dist.all_reduce(t); assert float(t[0]) == 1.0\n"
        "print('C_RCCL_WORLD1_OK')\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "C_RCCL_WORLD1_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_pipeline_does_not_trust_the_callers_end_header_hash():
    """ctx.end_header_hash is an OUTPUT of builder.skip (header_range.rs:42-55), not an input of the proof request: the pipeline
    sets it from the target header it hashed.  Garbage in the uploaded ranges' end hashes must change nothing (world 1)."""
    from blobstreamx_amd.engine import PipelinedEngines
    J, B, V, R = 8, 32, 20, 4
    w = synth.Workload(52, R, J, B, v=V, n_blocks=201)
    good = w.ranges.copy()
    w.ranges["end_header_hash"] = np.random.default_rng(1).integers(0, 256, size=w.ranges["end_header_hash"].shape, dtype=np.uint8)
    for kw in (dict(with_witness=True), dict(with_witness=False, with_commit=False)):
        pe = PipelinedEngines(J, B, V, R, n_engines=2, **kw)
        pe.upload_workload(w)
        pe.step(); pe.step()
        res = pe.download()
        assert not res["range_status"].any() and res["header_status"] == 0
        for r in range(R):
            rc, out, _, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])
            assert rc == T.OK
            if kw.get("with_commit", True):
                assert res["output64"][r].tobytes() == out, r
            else:
                assert res["output64"][r, 32:].tobytes() == out[32:], r          # the commitment; no skip -> first half is ctx.end_header_hash
                assert res["output64"][r, :32].tobytes() == bytes(good[r]["end_header_hash"]), r


@pytest.mark.parametrize("J,B,V,R,n_engines,n_sets,witness", [(8, 32, 20, 4, 2, 1, True), (32, 64, 100, 2, 2, 1, True), (8, 32, 20, 3, 1, 3, False)])
def test_pipeline_autotune_keeps_results(J, B, V, R, n_engines, n_sets, witness):
    """bsx_pipeline_autotune moves the chunks' main / side streams over the pipeline's stream pool between joined trials: the
    steps it runs are real steps, so after it (and after further un-joined steps on the assignment it kept) every range must
    still equal the oracle's, tampered range included; it fails before an upload."""
    from blobstreamx_amd import _lib
    from blobstreamx_amd.engine import Pipeline
    w = synth.Workload(7, R, J, B, v=V)
    w.validators[R - 1, 2]["signature"][3] ^= 4
    p = Pipeline(J, B, V, R, n_chunks=n_engines, n_sets=n_sets, with_witness=witness)
    with pytest.raises(_lib.BsxError) as e:
        p.autotune()
    assert "before bsx_pipeline_upload" in str(e.value)
    p.upload_workload(w)
    p.step()
    tune = p.autotune(2)
    hot = 2 * n_engines * n_sets
    assert tune["n_trials"] == 2 * (17 - hot) and 0 < tune["best_ms"] and 0 < tune["initial_ms"] <= tune["worst_ms"]
    assert len(set(tune["assignment"])) == hot and max(tune["assignment"]) < 16
    for _ in range(3):
        p.step()
    if witness:
        res = _check_pipelined_against_oracle(p, w, J, B)
    else:
        res = p.download()
        for r in range(R):
            rc, out, _, _ = oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])
            mine = res["skip_status"][r] if res["skip_status"][r] else (T.ERR_ASSERT if res["range_status"][r] else T.OK)
            assert mine == rc and res["output64"][r].tobytes() == out, r
    assert res["skip_status"][R - 1] == T.ERR_BAD_SIGNATURE and not res["skip_status"][:R - 1].any()


def test_pipeline_argument_errors():
    """bsx_pipeline_* report misuse with a status code and a message, never abort: bad shapes, unknown flags, a step before the
    upload, a multi-GPU step without the all-gather callback, inputs that do not cover the rank's slice, a target outside the
    supplied headers, unknown buffer ids."""
    import ctypes as C
    from blobstreamx_amd import _lib
    from blobstreamx_amd import engine as E
    with pytest.raises(_lib.BsxError) as e:
        E.Pipeline(3, 8, 4, 2)                                         # NB_MAP_JOBS not a power of two
    assert e.value.status == T.ERR_BAD_ARG
    with pytest.raises(_lib.BsxError):
        E.Pipeline(4, 8, 4, 3, n_chunks=2)                             # chunks do not divide the ranges
    with pytest.raises(_lib.BsxError):
        E.Pipeline(4, 8, 600, 2)                                       # v_max beyond what one tally workgroup folds
    with pytest.raises(_lib.BsxError):
        E.Pipeline(4, 8, 4, 2, subchain_form=9)
    with pytest.raises(_lib.BsxError):
        E.Pipeline(4, 8, 4, 2, rank=0, world=3)                        # world must divide the map jobs
    with pytest.raises(_lib.BsxError):
        E.Pipeline(4, 8, 4, 2, rank=2, world=2)                        # rank out of range
    L = _lib.lib()
    cfg = E._Config(4, 8, 4, 2, 1, 0, 1, 1 << 20, 0, 0, 0, (C.c_uint8 * 52)(), 0, 0, 1, 0)
    h = C.c_void_p()
    assert L.bsx_pipeline_create(_lib.context(0), C.byref(cfg), C.byref(h)) == T.ERR_BAD_ARG and b"unknown flags" in L.bsx_last_error()
    p = E.Pipeline(4, 8, 4, 2)
    with pytest.raises(_lib.BsxError) as e:
        p.step()
    assert "before bsx_pipeline_upload" in str(e.value)
    w = synth.Workload(3, 2, 4, 8, v=4)
    with pytest.raises(_lib.BsxError):
        p.upload(w.headers[:, :20], w.ranges, w.latest, w.validators, w.trusted)      # 20 headers per range < J*B + 1
    bad = w.ranges.copy()
    bad["end_block"][1] = bad["start_block"][1] + 40                                  # target header beyond the 33 supplied
    with pytest.raises(_lib.BsxError):
        p.upload(w.headers, bad, w.latest, w.validators, w.trusted)
    with pytest.raises(_lib.BsxError):
        p.upload(w.headers, w.ranges, w.latest)                                       # BSX_PIPE_COMMIT needs the validator sets
    with pytest.raises(_lib.BsxError):
        p.buffer(0, 99)
    with pytest.raises(_lib.BsxError):
        p.buffer(5, E.BUF_COMPACT)
    p.upload_workload(w)
    p.step()
    assert not p.download()["range_status"].any()
    p2 = E.Pipeline(4, 8, 4, 1, rank=1, world=2)                                      # no process group: no callback registered
    p2.upload_workload(synth.Workload(3, 2, 4, 8, v=4))
    with pytest.raises(_lib.BsxError) as e:
        p2.step()
    assert "bsx_pipeline_set_allgather" in str(e.value)
