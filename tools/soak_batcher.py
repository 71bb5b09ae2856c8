#!/usr/bin/env python3
"""Randomised soak of the coalescing front end against the oracle (a tool beside the suite: `python tools/soak_batcher.py 120 [seed]`).

Every round: a random circuit shape (J, B, V), a random number of submitter threads, a workload whose ranges carry DRIFTING validator
sets (synth rotate_permille) and random tampering (signature, chain link, data hash, voting power, trusted hash, malformed header,
foreign chain); the threads push a random mix of request kinds through ONE batcher in random order and with random pauses —
header_range (submit + wait), the per-map-job hint, hint-then-prove_subchain, the map closure as one request (bsx_map_job form) — and
every result is compared with the oracle's: status, 64-byte output, commit result, proofs, expected commitment, subchain record.
Batches of every composition occur (good and bad requests side by side, kinds on their own lanes, sets changing inside a batch)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import batcher as BT
from blobstreamx_amd import types as T

TAMPER = ["ok", "ok", "ok", "sig", "link", "data_hash", "power", "trusted_hash", "bad_header", "chain"]


def tamper(w, r, kind, rng):
    n = w.n_blocks
    if kind == "sig":
        signed = np.nonzero(w.validators[r]["is_signed"])[0]
        if len(signed):
            w.validators[r, int(rng.choice(signed))]["signature"][int(rng.integers(0, 64))] ^= 1 << int(rng.integers(0, 8))
    elif kind == "link":
        w.headers[r, int(rng.integers(1, max(2, n)))]["proposer"][5] ^= 1
    elif kind == "data_hash":
        w.headers[r, int(rng.integers(1, max(2, n)))]["hash"][1][7] ^= 0x80
    elif kind == "power":
        for v in range(w.validators.shape[1] // 2 + 1):
            w.validators[r, v]["is_signed"] = 0
    elif kind == "trusted_hash":
        w.ranges[r]["start_header_hash"][int(rng.integers(0, 32))] ^= 2
    elif kind == "bad_header":
        w.headers[r, int(rng.integers(0, n + 1))]["len"][3] = 60
    elif kind == "chain":
        w.headers[r, n]["chain_id"][3] ^= 1


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
    t_end = time.time() + budget
    rounds = n_range = n_hint = n_job = 0
    statuses = {}
    while time.time() < t_end:
        J = int(rng.choice([1, 2, 4, 8, 32]))
        B = int(rng.choice([2, 8, 16, 64]))
        V = int(rng.choice([1, 4, 20, 100]))
        R = int(rng.integers(1, 13))
        NT = int(rng.integers(1, 9))
        n_blocks = int(rng.integers(max(1, J * B // 2), J * B + 1))
        w = synth.Workload(int(rng.integers(1, 1 << 30)), R, J, B, v=V, n_blocks=n_blocks, rotate_permille=int(rng.choice([0, 0, 30, 300, 1000])),
                           nil_permille=int(rng.choice([0, 0, 60])), absent_permille=int(rng.choice([0, 0, 40])))
        kinds = [str(rng.choice(TAMPER)) for _ in range(R)]
        for r, k in enumerate(kinds):
            tamper(w, r, k, rng)
        want_range = [oracle.header_range(J, B, w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r])[:3]
                      for r in range(R)]
        bt = BT.Batcher(J, B, V, max_requests=int(rng.choice([4, 16, 32])), window_us=int(rng.choice([0, 20, 200])), n_lanes=int(rng.choice([0, 1, 2, 4])))
        # the request list: (kind, range, job)
        reqs = []
        for r in range(R):
            reqs.append(("range", r, 0))
            if kinds[r] == "bad_header":                  # the oracle's helpers below refuse a malformed header outright: the range request covers it
                continue
            for j in rng.choice(J, size=min(J, int(rng.integers(0, 4))), replace=False):
                reqs.append((str(rng.choice(["hint", "hint+sub", "job"])), r, int(j)))
        order = rng.permutation(len(reqs))
        errors = []
        go = threading.Barrier(NT)
        pauses = rng.random(len(reqs)) * 2e-4
        forms, hows = rng.integers(0, 3, len(reqs)), rng.integers(0, 3, len(reqs))
        packed = [BT.pack_headers(w.headers[r]) if kinds[r] != "bad_header" else None for r in range(R)]

        def check_range(r, rc, out, res, msg):
            wrc, wout, wres = want_range[r]
            assert rc == wrc, ("range", r, kinds[r], rc, wrc, msg)
            if rc in (T.OK, T.ERR_ASSERT, T.ERR_BAD_SIGNATURE, T.ERR_VOTING_POWER):
                assert out == wout and res.tobytes() == wres.tobytes(), ("range", r, kinds[r])

        hh_end = [oracle.header_hash_only(w.headers[r])[n_blocks].tobytes() if kinds[r] != "bad_header" else None for r in range(R)]

        def worker(t):
            try:
                go.wait()
                for i in order[t::NT]:
                    kind, r, j = reqs[int(i)]
                    time.sleep(float(pauses[int(i)]))
                    if kind == "range":
                        # round 6: a random upload form (512-byte records / packed wire headers / records that "stay" — pageable here, so
                        # staged all the same) and a random way to collect the result (wait / poll until done, then wait / wait twice)
                        form = int(forms[int(i)])
                        if form == 1 and kinds[r] != "bad_header":
                            tk = bt.submit_header_range(w.input48(r), packed[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r], packed=True)
                        else:
                            tk = bt.submit_header_range(w.input48(r), w.headers[r], int(w.first_height[r]), int(w.latest[r]), w.validators[r], w.trusted[r],
                                                        inputs_stay=(form == 2))
                        how = int(hows[int(i)])
                        if how == 1:
                            while not bt.done(tk):
                                time.sleep(2e-5)
                        rc, (out, res) = bt.wait(tk, allow=tuple(range(1, 10)))
                        if how == 2:
                            rc_again, (out2, _) = bt.wait(tk, allow=tuple(range(1, 10)))
                            assert rc_again == rc and out2 == out, ("second wait", r)
                        check_range(r, rc, out, res.copy(), _lib.last_error() if rc else "")
                        continue
                    S, latest = int(w.first_height[r]), int(w.latest[r])
                    E = S + n_blocks
                    bs, be = S + j * B, S + (j + 1) * B
                    hdr = w.headers[r][j * B:(j + 1) * B + 1]
                    orc, oh = oracle.data_commitment_inputs(hdr, bs, latest, bs, be, B)
                    wrc, wrec, _ = (oracle.prove_subchain(B, oh["start_header"], oh["end_header"], oh["data_hash_proofs"], oh["last_block_id_proofs"], bs, be, E,
                                                          hh_end[r]) if orc == T.OK else (orc, None, None))
                    if kind == "job":
                        rg = np.zeros(1, T.SHARED_CTX)
                        rg["start_block"], rg["end_block"] = S, E
                        rg["start_header_hash"][0] = np.frombuffer(oracle.header_hash_only(w.headers[r][:1])[0].tobytes(), np.uint8)
                        rg["end_header_hash"][0] = np.frombuffer(hh_end[r], np.uint8)
                        rc, got = bt.wait(bt.submit_map_job(rg, j, hdr, bs, latest), allow=tuple(range(1, 10)))
                        if orc != T.OK:
                            assert rc == orc, (kind, r, j, kinds[r], rc, orc)
                            continue
                        assert rc == wrc and got["record"].tobytes() == wrec.tobytes(), (kind, r, j, kinds[r], rc, wrc)
                    else:
                        rc, got = bt.wait(bt.submit_data_commitment_inputs(hdr, bs, latest, bs, be), allow=tuple(range(1, 10)))
                        assert rc == orc, (kind, r, j, kinds[r], rc, orc, _lib.last_error() if rc else "")
                        if orc != T.OK:
                            continue
                        assert got["expected_data_commitment"] == oh["expected_data_commitment"], (kind, r, j)
                    assert (got["start_header_hash"], got["end_header_hash"]) == (oh["start_header"], oh["end_header"]), (kind, r, j, kinds[r])
                    assert got["data_hash_proofs"].tobytes() == oh["data_hash_proofs"].tobytes(), (kind, r, j)
                    assert got["last_block_id_proofs"].tobytes() == oh["last_block_id_proofs"].tobytes(), (kind, r, j)
                    if kind == "hint+sub":
                        tk = bt.submit_prove_subchain(got["start_header_hash"], got["end_header_hash"], got["data_hash_proofs"], got["last_block_id_proofs"], bs, be, E,
                                                      hh_end[r])
                        rc2, rec = bt.wait(tk, allow=tuple(range(1, 10)))
                        assert rc2 == wrc and rec.tobytes() == wrec.tobytes(), (kind, r, j, kinds[r], rc2, wrc)
            except Exception as e:      # noqa: BLE001 — reported with the round's parameters
                import traceback
                errors.append(traceback.format_exc())

        th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        st = bt.stats()
        bt.close()
        if errors:
            print("FAILED round", rounds, dict(J=J, B=B, V=V, R=R, NT=NT, n_blocks=n_blocks, kinds=kinds))
            print(errors[0])
            return 1
        rounds += 1
        n_range += R
        n_hint += sum(1 for k, _, _ in reqs if k in ("hint", "hint+sub"))
        n_job += sum(1 for k, _, _ in reqs if k == "job")
        for rc, _, _ in want_range:
            statuses[rc] = statuses.get(rc, 0) + 1
        if rounds % 20 == 0:
            print("round %d: J=%d B=%d V=%d R=%d threads=%d sets %s ok" % (rounds, J, B, V, R, NT, {k: st[k]["batches"] for k in st}), flush=True)
    print("soak ok: %d rounds, %d header_range requests (statuses %s), %d hints, %d map jobs through the batcher compared with the oracle"
          % (rounds, n_range, {T.STATUS_NAMES[k]: v for k, v in sorted(statuses.items())}, n_hint, n_job))
    return 0


if __name__ == "__main__":
    sys.exit(main())
