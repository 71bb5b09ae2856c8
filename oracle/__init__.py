"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (blobstreamx_amd) never does.  See oracle/orc.h for the pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from blobstreamx_amd import types as T

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_SO_OVERRIDE") or os.path.join(_DIR, "liboracle.so")   # override: the sanitizer build
_lib = None


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h"))]
    srcs += [os.path.join(_DIR, "..", "include", f) for f in ("bsx.h", "bsx_layout.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-C", _DIR, "-B", "liboracle.so"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("ORACLE_SO_OVERRIDE"):
            build()            # mtime check: a library older than its sources is rebuilt (a stale one once overran a buffer sized by newer code)
        _lib = C.CDLL(_SO)
        _lib.orc_sha256_has_shani.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None  # data_as keeps `a` alive


def _b(x, n=None):
    a = np.frombuffer(bytes(x), dtype=np.uint8).copy()
    assert n is None or a.size == n, (a.size, n)
    return a


def sha256(msg):
    out = np.zeros(32, np.uint8)
    m = _b(msg) if len(msg) else np.zeros(1, np.uint8)
    lib().orc_sha256(_p(m), C.c_size_t(len(msg)), _p(out))
    return out.tobytes()


def sha256_force_portable(on):
    lib().orc_sha256_force_portable(C.c_int(int(on)))


def has_shani():
    return bool(lib().orc_sha256_has_shani())


def sha512(msg):
    out = np.zeros(64, np.uint8)
    m = _b(msg) if len(msg) else np.zeros(1, np.uint8)
    lib().orc_sha512(_p(m), C.c_size_t(len(msg)), _p(out))
    return out.tobytes()


def sc_reduce64(x):
    out = np.zeros(32, np.uint8)
    lib().orc_sc_reduce64(_p(_b(x, 64)), _p(out))
    return out.tobytes()


def ed25519_verify(pk, msg, sig):
    m = _b(msg) if len(msg) else np.zeros(1, np.uint8)
    return bool(lib().orc_ed25519_verify(_p(_b(pk, 32)), _p(m), C.c_size_t(len(msg)), _p(_b(sig, 64))))


def ed25519_verify_h(pk, sig, h):
    return bool(lib().orc_ed25519_verify_h(_p(_b(pk, 32)), _p(_b(sig, 64)), _p(_b(h, 32))))


def leaf_hash(x):
    out = np.zeros(32, np.uint8)
    lib().orc_leaf_hash(_p(_b(x) if len(x) else np.zeros(1, np.uint8)), C.c_size_t(len(x)), _p(out))
    return out.tobytes()


def inner_hash(l, r):
    out = np.zeros(32, np.uint8)
    lib().orc_inner_hash(_p(_b(l, 32)), _p(_b(r, 32)), _p(out))
    return out.tobytes()


def header_hashes(headers):
    """headers: ndarray[HEADER] -> (hashes[n,32], dh[n] DH_PROOF, lb[n] LB_PROOF)"""
    headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
    n = headers.size
    hashes = np.zeros((n, 32), np.uint8)
    dh = np.zeros(n, T.DH_PROOF)
    lb = np.zeros(n, T.LB_PROOF)
    for i in range(n):
        rc = lib().orc_header_hash(C.c_void_p(headers.ctypes.data + 512 * i), C.c_void_p(hashes.ctypes.data + 32 * i),
                                   C.c_void_p(dh.ctypes.data + 162 * i), C.c_void_p(lb.ctypes.data + 200 * i))
        if rc:
            raise ValueError(f"orc_header_hash: {T.STATUS_NAMES[rc]} at header {i}")
    return hashes, dh, lb


def header_hash_only(headers):
    headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
    hashes = np.zeros((headers.size, 32), np.uint8)
    for i in range(headers.size):
        rc = lib().orc_header_hash(C.c_void_p(headers.ctypes.data + 512 * i), C.c_void_p(hashes.ctypes.data + 32 * i),
                                   None, None)
        if rc:
            raise ValueError(f"orc_header_hash: {T.STATUS_NAMES[rc]} at header {i}")
    return hashes


def encode_data_root_tuple(data_hash, height):
    out = np.zeros(64, np.uint8)
    lib().orc_encode_data_root_tuple(_p(_b(data_hash, 32)), C.c_uint64(height), _p(out))
    return out.tobytes()


def get_data_commitment(data_hashes, start_block, end_block):
    dhs = np.ascontiguousarray(data_hashes, np.uint8).reshape(-1, 32)
    out = np.zeros(32, np.uint8)
    af = C.c_uint32(0)
    rc = lib().orc_get_data_commitment(_p(dhs), C.c_uint32(dhs.shape[0]), C.c_uint64(start_block), C.c_uint64(end_block),
                                       _p(out), C.byref(af))
    return rc, out.tobytes(), af.value


def data_commitment_inputs(headers, first_height, latest_block, start_block, end_block, max_leaves):
    headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
    sh, eh, exp = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(32, np.uint8)
    dh = np.zeros(max_leaves, T.DH_PROOF)
    lb = np.zeros(max_leaves, T.LB_PROOF)
    rc = lib().orc_data_commitment_inputs(_p(headers), C.c_uint64(first_height), C.c_uint64(headers.size),
                                          C.c_uint64(latest_block), C.c_uint64(start_block), C.c_uint64(end_block),
                                          C.c_uint32(max_leaves), _p(sh), _p(eh), _p(dh), _p(lb), _p(exp))
    return rc, dict(start_header=sh.tobytes(), end_header=eh.tobytes(), data_hash_proofs=dh, last_block_id_proofs=lb,
                    expected_data_commitment=exp.tobytes())


def prove_subchain(batch_size, start_header, end_header, dh, lb, batch_start, batch_end, global_end, global_end_hash,
                   ctx=None, want_witness=False):
    lay = T.map_layout(batch_size)
    rec = np.zeros(1, T.SUBCHAIN)
    dh = np.ascontiguousarray(dh, T.DH_PROOF)
    lb = np.ascontiguousarray(lb, T.LB_PROOF)
    compact = np.zeros(int(lay["compact_stride"]), np.uint8) if want_witness else None
    rc = lib().orc_prove_subchain(C.c_uint32(batch_size), _p(ctx) if ctx is not None else None, _p(_b(start_header, 32)),
                                  _p(_b(end_header, 32)), _p(dh), _p(lb), C.c_uint64(batch_start), C.c_uint64(batch_end),
                                  C.c_uint64(global_end), _p(_b(global_end_hash, 32)), _p(rec), _p(compact))
    return rc, rec[0], compact


def reduce(records):
    records = np.ascontiguousarray(records, T.SUBCHAIN)
    out = np.zeros(1, T.SUBCHAIN)
    n = records.size
    lay = T.reduce_layout()
    compact = np.zeros(max(n - 1, 1) * int(lay["compact_stride"]), np.uint8)
    rc = lib().orc_reduce(_p(records), C.c_uint32(n), _p(out), _p(compact))
    return rc, out[0], compact


def make_ctx(start_block, start_header_hash, end_block, end_header_hash):
    ctx = np.zeros(1, T.SHARED_CTX)
    ctx["start_block"], ctx["end_block"] = start_block, end_block
    ctx["start_header_hash"][0] = np.frombuffer(bytes(start_header_hash), np.uint8)
    ctx["end_header_hash"][0] = np.frombuffer(bytes(end_header_hash), np.uint8)
    return ctx


def prove_data_commitment(nb_map_jobs, batch_size, ctx, headers, first_height, latest_block, want_witness=False):
    headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
    ml, rl = T.map_layout(batch_size), T.reduce_layout()
    out = np.zeros(32, np.uint8)
    result = np.zeros(1, T.SUBCHAIN)
    records = np.zeros(nb_map_jobs, T.SUBCHAIN)
    csz = nb_map_jobs * int(ml["compact_stride"]) + (nb_map_jobs - 1) * int(rl["compact_stride"])
    compact = np.zeros(max(csz, 1), np.uint8) if want_witness else None
    st = C.c_uint32(0)
    rc = lib().orc_prove_data_commitment(C.c_uint32(nb_map_jobs), C.c_uint32(batch_size), _p(ctx), _p(headers),
                                         C.c_uint64(first_height), C.c_uint64(headers.size), C.c_uint64(latest_block),
                                         _p(out), _p(result), _p(records), _p(compact), C.byref(st))
    return rc, dict(data_commitment=out.tobytes(), result=result[0], records=records, compact=compact, status=st.value)


def prove_next_header_data_commitment(prev_block, prev_header_hash, next_block, header, latest_block):
    header = np.ascontiguousarray(header, dtype=T.HEADER).reshape(-1)
    out = np.zeros(32, np.uint8)
    rc = lib().orc_prove_next_header_data_commitment(C.c_uint64(prev_block), _p(_b(prev_header_hash, 32)),
                                                     C.c_uint64(next_block), _p(header), C.c_uint64(latest_block), _p(out))
    return rc, out.tobytes()


def expand_witness(layout, n_jobs, compact):
    layout = np.ascontiguousarray(layout, T.WITNESS_LAYOUT).reshape(1)
    out = np.zeros(n_jobs * int(layout["n_elements"][0]), np.uint64)
    lib().orc_expand_witness(_p(layout), C.c_uint32(n_jobs), _p(compact), _p(out))
    return out


def expand_range_witness(nb_map_jobs, batch_size, compact, v_max=None):
    """compact = jobs then reduce nodes (orc_prove_data_commitment) [then the COMMIT and SKIP units (orc_header_range) when
    v_max is given] -> expanded u64 (same order)"""
    ml, rl = T.map_layout(batch_size), T.reduce_layout()
    a = expand_witness(ml, nb_map_jobs, compact)
    off = nb_map_jobs * int(ml["compact_stride"])
    b = expand_witness(rl, nb_map_jobs - 1, compact[off:]) if nb_map_jobs > 1 else np.zeros(0, np.uint64)
    parts = [a, b]
    if v_max is not None:
        off += (nb_map_jobs - 1) * int(rl["compact_stride"])
        cl, sl = T.commit_layout(v_max), T.skip_layout(v_max)
        parts.append(expand_witness(cl, 1, compact[off:]))
        parts.append(expand_witness(sl, 1, compact[off + int(cl["compact_stride"]):]))
    return np.concatenate(parts)


def range_compact_bytes(nb_map_jobs, batch_size, v_max):
    """bytes of the compact witness orc_header_range writes: map jobs, reduce nodes, COMMIT unit, SKIP unit"""
    return (nb_map_jobs * int(T.map_layout(batch_size)["compact_stride"]) + (nb_map_jobs - 1) * int(T.reduce_layout()["compact_stride"])
            + int(T.commit_layout(v_max)["compact_stride"]) + int(T.skip_layout(v_max)["compact_stride"]))


def sha512_challenge(validators):
    validators = np.ascontiguousarray(validators, T.VALIDATOR).reshape(-1)
    h = np.zeros((validators.size, 32), np.uint8)
    dig = np.zeros((validators.size, 64), np.uint8)
    for i in range(validators.size):
        lib().orc_sha512_challenge(C.c_void_p(validators.ctypes.data + 256 * i), C.c_void_p(h.ctypes.data + 32 * i),
                                   C.c_void_p(dig.ctypes.data + 64 * i))
    return h, dig


def verify_commit(validators, header_hash, want_witness=False):
    """-> (result, sig_ok[, compact COMMIT unit when want_witness])"""
    validators = np.ascontiguousarray(validators, T.VALIDATOR).reshape(-1)
    res = np.zeros(1, T.COMMIT_RESULT)
    ok = np.zeros(validators.size, np.uint8)
    cw = np.zeros(int(T.commit_layout(validators.size)["compact_stride"]), np.uint8) if want_witness else None
    lib().orc_verify_commit_w(_p(validators), C.c_uint32(validators.size), _p(_b(header_hash, 32)), _p(res), _p(ok), _p(cw))
    return (res[0], ok, cw) if want_witness else (res[0], ok)


CHAIN_ID = b"celestia"      # the synthetic workload's chain (synth.CHAIN_ID); the mocha-4 fixtures pass b"mocha-4"


def header_range(nb_map_jobs, batch_size, input48, headers, first_height, latest_block, target_validators,
                 trusted_validators, want_witness=False, chain_id=CHAIN_ID):
    headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
    tv = np.ascontiguousarray(target_validators, T.VALIDATOR).reshape(-1)
    rv = np.ascontiguousarray(trusted_validators, T.VALIDATOR).reshape(-1)
    assert tv.size == rv.size
    compact = np.zeros(range_compact_bytes(nb_map_jobs, batch_size, tv.size), np.uint8) if want_witness else None
    out = np.zeros(64, np.uint8)
    res = np.zeros(1, T.COMMIT_RESULT)
    rc = lib().orc_header_range(C.c_uint32(nb_map_jobs), C.c_uint32(batch_size), _p(_b(input48, 48)), _p(headers),
                                C.c_uint64(first_height), C.c_uint64(headers.size), C.c_uint64(latest_block), _p(tv), _p(rv),
                                C.c_uint32(tv.size), _p(_b(chain_id)), C.c_uint32(len(chain_id)), _p(out), _p(res), _p(compact))
    return rc, out.tobytes(), res[0], compact


def next_header(input40, prev_header, next_header_, latest_block, next_validators, chain_id=CHAIN_ID, want_witness=False):
    """CombinedStepCircuit::define (circuits/next_header.rs:25-46) -> (rc, output64, commit_result[, compact COMMIT + STEP units])."""
    ph = np.ascontiguousarray(prev_header, T.HEADER).reshape(1)
    nh = np.ascontiguousarray(next_header_, T.HEADER).reshape(1)
    nv = np.ascontiguousarray(next_validators, T.VALIDATOR).reshape(-1)
    out = np.zeros(64, np.uint8)
    res = np.zeros(1, T.COMMIT_RESULT)
    cw = None
    if want_witness:
        cw = np.zeros(int(T.commit_layout(nv.size)["compact_stride"]) + int(T.step_layout()["compact_stride"]), np.uint8)
    rc = lib().orc_next_header_w(_p(_b(input40, 40)), _p(ph), _p(nh), C.c_uint64(latest_block), _p(nv), C.c_uint32(nv.size),
                                 _p(_b(chain_id)), C.c_uint32(len(chain_id)), _p(out), _p(res), _p(cw))
    return (rc, out.tobytes(), res[0], cw) if want_witness else (rc, out.tobytes(), res[0])


def expand_step_witness(v_max, compact):
    """compact COMMIT + STEP units (next_header) -> expanded u64"""
    cl, tl = T.commit_layout(v_max), T.step_layout()
    return np.concatenate([expand_witness(cl, 1, compact), expand_witness(tl, 1, compact[int(cl["compact_stride"]):])])


SKIP_EVAL = np.dtype([("overlap_power", "<u8"), ("start_total_power", "<u8"), ("signed_power", "<u8"),
                      ("target_total_power", "<u8"), ("valid", "<u4"), ("power_overflow", "<u4")])


def find_block_to_request(start_block, max_end_block, start_validators, candidate_heights, candidate_validators):
    """fetcher.rs:60-87 over pre-fetched candidates -> (rc, block, evals[n_candidates])."""
    sv = np.ascontiguousarray(start_validators, T.VALIDATOR).reshape(-1)
    cv = np.ascontiguousarray(candidate_validators, T.VALIDATOR).reshape(-1, sv.size)
    hs = np.ascontiguousarray(candidate_heights, np.uint64)
    assert cv.shape[0] == hs.size
    ev = np.zeros(hs.size, SKIP_EVAL)
    out = C.c_uint64(0)
    rc = lib().orc_find_block_to_request(C.c_uint64(start_block), C.c_uint64(max_end_block), _p(sv), C.c_uint32(hs.size), _p(hs),
                                         _p(cv), C.c_uint32(sv.size), C.byref(out), _p(ev))
    return rc, int(out.value), ev


def bench_header_range(nb_map_jobs, batch_size, ranges, headers, headers_per_range, latest, target, trusted, v_max,
                       with_witness, n_threads, reps=1, chain_id=CHAIN_ID):
    ranges = np.ascontiguousarray(ranges, T.SHARED_CTX).reshape(-1)
    n = ranges.size
    out64 = np.zeros((n, 64), np.uint8)
    cs = C.c_uint64(0)
    latest = np.ascontiguousarray(latest, np.uint64)
    rc = lib().orc_bench_header_range(C.c_uint32(n), C.c_uint32(reps), C.c_uint32(nb_map_jobs), C.c_uint32(batch_size), _p(ranges),
                                      _p(headers), C.c_uint64(headers_per_range), _p(latest), _p(target), _p(trusted),
                                      C.c_uint32(v_max), _p(_b(chain_id)), C.c_uint32(len(chain_id)), C.c_int(int(with_witness)),
                                      C.c_int(n_threads), _p(out64), C.byref(cs))
    return rc, out64, cs.value


# ---------------------------------------------------------------- Poseidon over Goldilocks (oracle/poseidon.c)
def poseidon_round_constants():
    lib().orc_poseidon_round_constants.restype = C.POINTER(C.c_uint64)
    ptr = lib().orc_poseidon_round_constants()
    return np.ctypeslib.as_array(ptr, shape=(360,)).copy()


def poseidon_permute(state):
    s = np.ascontiguousarray(state, np.uint64).reshape(12).copy()
    lib().orc_poseidon_permute(_p(s))
    return s


def poseidon_permute_fast(state):
    """the cpu_baseline form of the permutation (oracle/poseidon.c): must equal poseidon_permute"""
    s = np.ascontiguousarray(state, np.uint64).reshape(12).copy()
    lib().orc_poseidon_permute_fast(_p(s))
    return s


def bench_witness_caps(layout, compact, n_jobs, leaf_len, n_leaves, cap_height, n_threads, reps=1):
    """cpu_baseline of the witness commitment: caps [n_jobs, 2^cap_height, 4] of n_jobs compact witnesses on n_threads threads"""
    lay = np.ascontiguousarray(layout, T.WITNESS_LAYOUT).reshape(1)
    c = np.ascontiguousarray(compact, np.uint8).reshape(-1)
    assert c.size >= n_jobs * int(lay["compact_stride"][0])
    caps = np.zeros((n_jobs, 1 << cap_height, 4), np.uint64)
    rc = lib().orc_bench_witness_caps(_p(lay), C.c_uint32(n_jobs), C.c_uint32(reps), _p(c), C.c_uint32(leaf_len), C.c_uint32(n_leaves),
                                      C.c_uint32(cap_height), C.c_int(n_threads), _p(caps))
    assert rc == 0, rc
    return caps


def poseidon_hash_no_pad(elems):
    e = np.ascontiguousarray(elems, np.uint64).reshape(-1)
    out = np.zeros(4, np.uint64)
    lib().orc_poseidon_hash_no_pad(_p(e if e.size else np.zeros(1, np.uint64)), C.c_uint64(e.size), _p(out))
    return out


def poseidon_hash_or_noop(elems):
    e = np.ascontiguousarray(elems, np.uint64).reshape(-1)
    out = np.zeros(4, np.uint64)
    lib().orc_poseidon_hash_or_noop(_p(e if e.size else np.zeros(1, np.uint64)), C.c_uint64(e.size), _p(out))
    return out


def poseidon_two_to_one(l, r):
    out = np.zeros(4, np.uint64)
    lib().orc_poseidon_two_to_one(_p(np.ascontiguousarray(l, np.uint64)), _p(np.ascontiguousarray(r, np.uint64)), _p(out))
    return out


def poseidon_tree_digests(n_leaves, cap_height):
    return 2 * n_leaves - (1 << cap_height)


def poseidon_merkle_tree(elements, leaf_len, n_leaves, cap_height):
    """MerkleTree::new over rows of leaf_len elements (zero padded to n_leaves rows) -> (tree [digests, 4], cap)."""
    e = np.ascontiguousarray(elements, np.uint64).reshape(-1)
    tree = np.zeros((poseidon_tree_digests(n_leaves, cap_height), 4), np.uint64)
    rc = lib().orc_poseidon_merkle_tree(_p(e), C.c_uint64(e.size), C.c_uint32(leaf_len), C.c_uint32(n_leaves),
                                        C.c_uint32(cap_height), _p(tree))
    assert rc == 0, rc
    return tree, tree[-(1 << cap_height):]


def commit_fold(results, first_index=0):
    """checker twin of bsx_dev_verify_commits' fold: COMMIT_RESULT[n] -> COMMIT_FOLD record."""
    r = np.ascontiguousarray(results, T.COMMIT_RESULT).reshape(-1)
    out = np.zeros(1, T.COMMIT_FOLD)
    lib().orc_commit_fold(_p(r), C.c_uint32(r.size), C.c_uint32(first_index), _p(out))
    return out[0]


def bench_verify_commits(validators, header_hashes, n_threads, reps=1):
    """mode S driver: validators [n_commits, v_max] VALIDATOR, header_hashes [n_commits, 32] -> (results, sig_ok)."""
    v = np.ascontiguousarray(validators, T.VALIDATOR)
    n, vmax = v.shape
    hh = np.ascontiguousarray(header_hashes, np.uint8).reshape(n, 32)
    res = np.zeros(n, T.COMMIT_RESULT)
    ok = np.zeros((n, vmax), np.uint8)
    rc = lib().orc_bench_verify_commits(C.c_uint32(n), C.c_uint32(reps), C.c_uint32(vmax), _p(v), _p(hh), C.c_int(n_threads), _p(res), _p(ok))
    assert rc == 0
    return res, ok
