"""Host-side mirror of the reference's builder / hint / circuit interface for the header_range path.

Same names, argument meaning and failure behaviour as the reference (panics / failed circuit assertions become
BsxError with the matching bsx_status), every call going through the C ABI of libbsx.so (include/bsx.h) — so the
parity tests read like the reference's own tests:

  DataCommitmentBuilder            trait DataCommitmentBuilder            circuits/builder.rs:20-79
    .encode_data_root_tuple        builder.rs:23-27,82-103
    .get_data_commitment           builder.rs:33-38,105-148
    .prove_subchain                builder.rs:45-52,150-271
    .prove_data_commitment         builder.rs:58-67,273-409
    .prove_next_header_data_commitment  builder.rs:73-78,411-443
  InputDataFetcher                 trait DataCommitmentInputFetcher       circuits/input.rs:39-61
    .get_data_commitment_inputs    input.rs:57-60,149-271  (= DataCommitmentOffchainInputs::hint, data_commitment.rs:18-45)
  CombinedSkipCircuit.prove        CombinedSkipCircuit::define            circuits/header_range.rs:32-59
  verify_commits                   inner loop of builder.skip/step        header_range.rs:42-48, next_header.rs:32-36
"""
import ctypes as C

import numpy as np

from . import _lib
from . import types as T


def _b(x, n):
    a = np.frombuffer(bytes(x), dtype=np.uint8).copy()
    if a.size != n:
        raise ValueError(f"expected {n} bytes, got {a.size}")
    return a


class InputDataFetcher:
    """The header source the hint reads (the reference's RPC / fixture fetcher, circuits/input.rs:66): a
    contiguous array of packed headers starting at `first_height`, and the chain head `latest_block`."""

    def __init__(self, headers, first_height, latest_block, device=0):
        self.headers = np.ascontiguousarray(headers, dtype=T.HEADER).reshape(-1)
        if self.headers.nbytes >= 1 << 16:
            # page-locked copy: the host tier uploads the headers by direct DMA instead of through the driver's staging buffer
            try:
                import torch
                self._pinned = torch.empty(self.headers.nbytes, dtype=torch.uint8, pin_memory=True)
                pin = self._pinned.numpy().view(T.HEADER)
                pin[:] = self.headers
                self.headers = pin
            except Exception:
                pass
        self.first_height = int(first_height)
        self.latest_block = int(latest_block)
        self.device = device

    def get_latest_block_number(self):            # input.rs:112-117
        return self.latest_block

    def header_hashes(self):
        n = self.headers.size
        hashes = np.zeros((n, 32), np.uint8)
        dh = np.zeros(n, T.DH_PROOF)
        lb = np.zeros(n, T.LB_PROOF)
        L = _lib.lib()
        _lib.check(L.bsx_header_hashes(_lib.context(self.device), _lib.p(self.headers), C.c_uint64(n), _lib.p(hashes), None, None))
        return hashes

    def get_inclusion_proofs(self):
        """hashes + (data_hash, last_block_id) inclusion proofs of every header (input.rs:175-179,188-195)."""
        n = self.headers.size
        hashes = np.zeros((n, 32), np.uint8)
        dh = np.zeros(n, T.DH_PROOF)
        lb = np.zeros(n, T.LB_PROOF)
        _lib.check(_lib.lib().bsx_header_hashes(_lib.context(self.device), _lib.p(self.headers), C.c_uint64(n), _lib.p(hashes),
                                                _lib.p(dh), _lib.p(lb)))
        return hashes, dh, lb

    def get_data_commitment_inputs(self, start_block_number, end_block_number, max_leaves):
        """circuits/input.rs:149-271 -> dict mirroring DataCommitmentInputs (input.rs:29-37)."""
        sh, eh, exp = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(32, np.uint8)
        dh = np.zeros(max_leaves, T.DH_PROOF)
        lb = np.zeros(max_leaves, T.LB_PROOF)
        _lib.check(_lib.lib().bsx_data_commitment_inputs(
            _lib.context(self.device), _lib.p(self.headers), C.c_uint64(self.first_height), C.c_uint64(self.headers.size),
            C.c_uint64(self.latest_block), C.c_uint64(start_block_number), C.c_uint64(end_block_number), C.c_uint32(max_leaves),
            _lib.p(sh), _lib.p(eh), _lib.p(dh), _lib.p(lb), _lib.p(exp)))
        return dict(start_header_hash=sh.tobytes(), end_header_hash=eh.tobytes(), data_hash_proofs=dh,
                    last_block_id_proofs=lb, expected_data_commitment=exp.tobytes())


class DataCommitmentBuilder:
    def __init__(self, device=0):
        self.device = device

    @property
    def _ctx(self):
        return _lib.context(self.device)

    def encode_data_root_tuple(self, data_hash, height):
        out = np.zeros(64, np.uint8)
        _lib.check(_lib.lib().bsx_encode_data_root_tuple(self._ctx, _lib.p(_b(data_hash, 32)), C.c_uint64(height), _lib.p(out)))
        return out.tobytes()

    def get_data_commitment(self, data_hashes, start_block, end_block):
        dhs = np.ascontiguousarray(data_hashes, np.uint8).reshape(-1, 32)
        out = np.zeros(32, np.uint8)
        _lib.check(_lib.lib().bsx_get_data_commitment(self._ctx, _lib.p(dhs), C.c_uint32(dhs.shape[0]), C.c_uint64(start_block),
                                                      C.c_uint64(end_block), _lib.p(out)))
        return out.tobytes()

    def prove_subchain(self, data_comm_proof, batch_start_block, batch_end_block, global_end_block, global_end_header_hash,
                       want_witness=False, raise_on_assert=True):
        """data_comm_proof: dict with start_header / end_header (or *_hash) + data_hash_proofs + last_block_id_proofs
        (DataCommitmentProofVariable, circuits/vars.rs:13-26).  Returns (record, witness | None)."""
        dh = np.ascontiguousarray(data_comm_proof["data_hash_proofs"], T.DH_PROOF)
        lb = np.ascontiguousarray(data_comm_proof["last_block_id_proofs"], T.LB_PROOF)
        B = dh.size
        sh = data_comm_proof.get("start_header", data_comm_proof.get("start_header_hash"))
        eh = data_comm_proof.get("end_header", data_comm_proof.get("end_header_hash"))
        rec = np.zeros(1, T.SUBCHAIN)
        wit = None
        if want_witness:
            lay = T.map_layout(B)
            wit = np.zeros(int(lay["n_elements"]), np.uint64)
        rc = _lib.lib().bsx_prove_subchain(self._ctx, C.c_uint32(B), _lib.p(_b(sh, 32)), _lib.p(_b(eh, 32)), _lib.p(dh), _lib.p(lb),
                                           C.c_uint64(batch_start_block), C.c_uint64(batch_end_block), C.c_uint64(global_end_block),
                                           _lib.p(_b(global_end_header_hash, 32)), _lib.p(rec), _lib.p(wit))
        _lib.check(rc, allow=() if raise_on_assert else (T.ERR_ASSERT,))
        return rec[0], wit

    def reduce(self, records):
        records = np.ascontiguousarray(records, T.SUBCHAIN)
        out = np.zeros(1, T.SUBCHAIN)
        _lib.check(_lib.lib().bsx_reduce(self._ctx, _lib.p(records), C.c_uint32(records.size), _lib.p(out)))
        return out[0]

    def prove_data_commitment(self, fetcher, nb_map_jobs, batch_size, start_block, start_header_hash, end_block, end_header_hash,
                              want_witness=False, raise_on_assert=True):
        """prove_data_commitment::<C, NB_MAP_JOBS, BATCH_SIZE> with the hint served by `fetcher`.
        Returns dict(data_commitment, result, records, witness, rc)."""
        ctx = np.zeros(1, T.SHARED_CTX)
        ctx["start_block"], ctx["end_block"] = start_block, end_block
        ctx["start_header_hash"][0] = _b(start_header_hash, 32)
        ctx["end_header_hash"][0] = _b(end_header_hash, 32)
        out = np.zeros(32, np.uint8)
        result = np.zeros(1, T.SUBCHAIN)
        records = np.zeros(nb_map_jobs, T.SUBCHAIN)
        wit = None
        if want_witness:
            ml, rl = T.map_layout(batch_size), T.reduce_layout()
            wit = np.zeros(nb_map_jobs * int(ml["n_elements"]) + (nb_map_jobs - 1) * int(rl["n_elements"]), np.uint64)
        rc = _lib.lib().bsx_prove_data_commitment(
            self._ctx, C.c_uint32(nb_map_jobs), C.c_uint32(batch_size), _lib.p(ctx), _lib.p(fetcher.headers),
            C.c_uint64(fetcher.first_height), C.c_uint64(fetcher.headers.size), C.c_uint64(fetcher.latest_block), _lib.p(out),
            _lib.p(result), _lib.p(records), _lib.p(wit))
        _lib.check(rc, allow=() if raise_on_assert else (T.ERR_ASSERT,))
        return dict(data_commitment=out.tobytes(), result=result[0], records=records, witness=wit, rc=rc)

    def prove_next_header_data_commitment(self, fetcher, prev_block_number, prev_header_hash, next_block_number):
        header = fetcher.headers[prev_block_number - fetcher.first_height:prev_block_number - fetcher.first_height + 1]
        out = np.zeros(32, np.uint8)
        _lib.check(_lib.lib().bsx_prove_next_header_data_commitment(
            self._ctx, C.c_uint64(prev_block_number), _lib.p(_b(prev_header_hash, 32)), C.c_uint64(next_block_number),
            _lib.p(np.ascontiguousarray(header)), C.c_uint64(fetcher.latest_block), _lib.p(out)))
        return out.tobytes()


def verify_commits(validators, header_hashes, device=0, want_witness=False):
    """validators: ndarray[n_commits, v_max] of VALIDATOR; header_hashes: [n_commits, 32] -> (results, sig_ok[, witness]).
    witness: u64 [n_commits, commit_layout(v_max).n_elements] — one COMMIT unit per commit (include/bsx_layout.h)."""
    v = np.ascontiguousarray(validators, T.VALIDATOR)
    if v.ndim == 1:
        v = v.reshape(1, -1)
    n, vmax = v.shape
    hh = np.ascontiguousarray(header_hashes, np.uint8).reshape(n, 32)
    res = np.zeros(n, T.COMMIT_RESULT)
    ok = np.zeros((n, vmax), np.uint8)
    wit = np.zeros((n, int(T.commit_layout(vmax)["n_elements"])), np.uint64) if want_witness else None
    _lib.check(_lib.lib().bsx_verify_commits_cap(_lib.context(device), _lib.p(v), C.c_uint32(n), C.c_uint32(vmax), _lib.p(hh),
                                                 _lib.p(res), _lib.p(ok), _lib.p(wit), C.c_uint64(wit.size if wit is not None else 0)))
    return (res, ok, wit) if want_witness else (res, ok)


class CombinedStepCircuit:
    """CombinedStepCircuit<MAX_VALIDATOR_SET_SIZE, CHAIN_ID_SIZE, C> (circuits/next_header.rs:11-46; instantiated with 100
    validators by bin/next_header{,_mocha}.rs:5-8)."""

    def __init__(self, max_validator_set_size, device=0, chain_id=b"celestia"):
        """chain_id = C::CHAIN_ID_BYTES, the circuit constant builder.step is called with (next_header.rs:32-33;
        circuits/config.rs:6-28: celestia / mocha-4)."""
        self.V, self.device, self.chain_id = max_validator_set_size, device, bytes(chain_id)

    def prove(self, input40, prev_header, next_header, latest_block, next_validators, want_witness=False, allow=()):
        """40-byte EVM-packed input (prev_block_number ‖ prev_header_hash) -> (64-byte output, commit result[, witness]).
        witness: u64 [next_header_witness_elements(V)] = the COMMIT unit of the next header's commit, then the STEP unit.
        allow: status codes that do not raise (then `self.last_rc` says which) — a failing request still has a witness."""
        if len(input40) != 40:
            raise ValueError("input must be 40 bytes: uint64 prev_block_number ‖ bytes32 prev_header_hash")
        ph = np.ascontiguousarray(prev_header, T.HEADER).reshape(1)
        nh = np.ascontiguousarray(next_header, T.HEADER).reshape(1)
        nv = np.ascontiguousarray(next_validators, T.VALIDATOR).reshape(-1)
        if nv.size != self.V:
            raise ValueError(f"validator array must have MAX_VALIDATOR_SET_SIZE = {self.V} slots")
        inp = np.frombuffer(bytes(input40), np.uint8).copy()
        out = np.zeros(64, np.uint8)
        res = np.zeros(1, T.COMMIT_RESULT)
        cid = np.frombuffer(self.chain_id, np.uint8).copy() if self.chain_id else None
        wit = np.zeros(T.next_header_witness_elements(self.V), np.uint64) if want_witness else None
        self.last_rc = _lib.check(_lib.lib().bsx_next_header_cap(_lib.context(self.device), _lib.p(inp), _lib.p(ph), _lib.p(nh), C.c_uint64(latest_block),
                                                                 _lib.p(nv), C.c_uint32(self.V), _lib.p(cid), C.c_uint32(len(self.chain_id)), _lib.p(out),
                                                                 _lib.p(res), _lib.p(wit), C.c_uint64(wit.size if wit is not None else 0)), allow=allow)
        return (out.tobytes(), res[0], wit) if want_witness else (out.tobytes(), res[0])


def find_block_to_request(start_block, max_end_block, start_validators, candidate_heights, candidate_validators, device=0):
    """The operator's skip-target search (circuits/fetcher.rs:60-87 `find_block_to_request`, called at
    bin/blobstreamx.rs:221-225) over pre-fetched candidates: returns (block, evals).  candidate_heights must contain
    every height of the halving sequence max_end, (max_end+start)/2, ... that the loop visits; `halving_sequence`
    lists them.  is_valid_skip is [UPSTREAM] (see include/bsx.h): parity unpinned."""
    sv = np.ascontiguousarray(start_validators, T.VALIDATOR).reshape(-1)
    hs = np.ascontiguousarray(candidate_heights, np.uint64)
    cv = np.ascontiguousarray(candidate_validators, T.VALIDATOR).reshape(hs.size, sv.size)
    ev = np.zeros(hs.size, T.SKIP_EVAL)
    out = C.c_uint64(0)
    _lib.check(_lib.lib().bsx_find_block_to_request(_lib.context(device), C.c_uint64(start_block), C.c_uint64(max_end_block),
                                                    _lib.p(sv), C.c_uint32(hs.size), _lib.p(hs), _lib.p(cv), C.c_uint32(sv.size),
                                                    C.byref(out), _lib.p(ev)))
    return int(out.value), ev


def halving_sequence(start_block, max_end_block):
    """Heights find_block_to_request can visit, in order (fetcher.rs:61-85)."""
    out, c = [], max_end_block
    while c - start_block > 1:
        out.append(c)
        c = (c + start_block) // 2
    return out


class CombinedSkipCircuit:
    """CombinedSkipCircuit<MAX_VALIDATOR_SET_SIZE, CHAIN_ID_SIZE, C, NB_MAP_JOBS, BATCH_SIZE>
    (circuits/header_range.rs:13-59; instantiated 100/32/32 and 100/32/64 by bin/header_range_{1024,2048}.rs:6-17)."""

    def __init__(self, max_validator_set_size, nb_map_jobs, batch_size, skip_max=None, device=0, chain_id=b"celestia"):
        """chain_id = C::CHAIN_ID_BYTES, the circuit constant builder.skip is called with (header_range.rs:42-43;
        circuits/config.rs:6-28: celestia / mocha-4)."""
        self.V, self.J, self.B = max_validator_set_size, nb_map_jobs, batch_size
        self.chain_id = bytes(chain_id)
        skip_max = skip_max if skip_max is not None else nb_map_jobs * batch_size
        # header_range.rs:37-40 (build-time assert)
        assert nb_map_jobs * batch_size <= skip_max, "NB_MAP_JOBS * BATCH_SIZE must be <= than SKIP_MAX"
        self.device = device

    def _witness_buffer(self, n):
        """Host buffer the witness is downloaded into: page-locked and kept for the circuit object, so that the 115 MB of a
        header_range_2048 witness arrive by direct DMA (≈ 50 GB/s) instead of through the driver's staging copy into freshly
        mapped pageable memory (≈ 10 GB/s: 12 ms -> 3 ms per proof).  Falls back to a plain array without torch / CUDA."""
        buf = getattr(self, "_wit_pinned", None)
        if buf is None or buf.numel() != n:
            try:
                import torch
                buf = torch.empty(n, dtype=torch.int64, pin_memory=True)
            except Exception:
                return np.zeros(n, np.uint64)
            self._wit_pinned = buf
        return buf.numpy().view(np.uint64)

    def prove(self, input48, fetcher, target_validators, trusted_validators, want_witness=False, allow=()):
        """48-byte EVM-packed input -> 64-byte output (target_header_hash ‖ data_commitment).  want_witness: also the
        Goldilocks witness of the WHOLE circuit (map jobs, reduce nodes, the COMMIT unit of the target commit, the SKIP unit;
        T.header_range_witness_elements), as a view of a page-locked buffer that the next prove() of this object overwrites
        (copy it to keep it).  allow: status codes that do not raise (`self.last_rc`)."""
        tv = np.ascontiguousarray(target_validators, T.VALIDATOR).reshape(-1)
        rv = np.ascontiguousarray(trusted_validators, T.VALIDATOR).reshape(-1)
        if tv.size != self.V or rv.size != self.V:
            raise ValueError(f"validator arrays must have MAX_VALIDATOR_SET_SIZE = {self.V} slots")
        out = np.zeros(64, np.uint8)
        res = np.zeros(1, T.COMMIT_RESULT)
        wit = None
        if want_witness:
            wit = self._witness_buffer(T.header_range_witness_elements(self.J, self.B, self.V))
        if getattr(self, "_cid", None) is None:         # per-object constants of the call, marshalled once
            self._cid = np.frombuffer(self.chain_id, np.uint8).copy() if self.chain_id else None
            self._fixed = (_lib.lib().bsx_header_range_cap, _lib.context(self.device), C.c_uint32(self.J), C.c_uint32(self.B), C.c_uint32(self.V),
                           _lib.p(self._cid), C.c_uint32(len(self.chain_id)))
        fn, ctx, cJ, cB, cV, cid, cidn = self._fixed
        self.last_rc = _lib.check(fn(
            ctx, cJ, cB, _lib.p(_b(input48, 48)), _lib.p(fetcher.headers),
            C.c_uint64(fetcher.first_height), C.c_uint64(fetcher.headers.size), C.c_uint64(fetcher.latest_block), _lib.p(tv),
            _lib.p(rv), cV, cid, cidn, _lib.p(out), _lib.p(res), _lib.p(wit), C.c_uint64(wit.size if wit is not None else 0)), allow=allow)
        return out.tobytes(), res[0], wit


def witness_manifest(batch_size):
    """bsx_witness_manifest -> ndarray[MANIFEST_ENTRY]: the variable groups of one map job's witness (batch_size = 0: of one
    reduce node) with their element offsets; `witness_view` slices a witness by group name."""
    return witness_manifest_section(T.SECTION_MAP if batch_size else T.SECTION_REDUCE, batch_size)


def witness_manifest_section(section, param=0):
    """bsx_witness_manifest_section -> ndarray[MANIFEST_ENTRY] of one unit: T.SECTION_MAP (param = BATCH_SIZE), SECTION_REDUCE,
    SECTION_COMMIT / SECTION_SKIP (param = validator slots), SECTION_STEP."""
    L = _lib.lib()
    n = C.c_uint32(0)
    _lib.check(L.bsx_witness_manifest_section(C.c_uint32(section), C.c_uint32(param), None, C.c_uint32(0), C.byref(n)))
    ent = np.zeros(n.value, T.MANIFEST_ENTRY)
    _lib.check(L.bsx_witness_manifest_section(C.c_uint32(section), C.c_uint32(param), _lib.p(ent), C.c_uint32(n.value), C.byref(n)))
    return ent


def witness_view(manifest, witness, name):
    """Elements of the group `name` of ONE job's witness -> [repeat, elements_per_record] (u64).  BYTES groups decode with
    np.packbits(v.astype(np.uint8), axis=1) (MSB first), U32 pairs as lo + (hi << 32)."""
    e = manifest[[m["name"].decode() == name or m["name"].decode().startswith(name + " ") for m in manifest]]
    if e.size != 1:
        raise KeyError(name)
    e = e[0]
    idx = int(e["element_offset"]) + np.arange(int(e["repeat"]))[:, None] * int(e["record_stride"]) + np.arange(int(e["elements_per_record"]))[None, :]
    return np.asarray(witness)[idx]
