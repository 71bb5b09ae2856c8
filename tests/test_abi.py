"""CPU suite: the C-ABI library builds/loads and exports every symbol include/bsx.h declares; without a GPU it
refuses to run (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from blobstreamx_amd import _lib
from blobstreamx_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_is_built_and_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "bsx.h")).read()
    declared = set(re.findall(r"\b(bsx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"bsx_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/bsx.h but not exported by libbsx.so"
    assert set(_lib.SYMBOLS) == declared
    assert L.bsx_version() == 0x00030200
    assert L.bsx_status_str(C.c_int(T.ERR_ASSERT)) == b"BSX_ERR_ASSERT"


def test_layout_queries_match_python_twin():
    L = _lib.lib()
    for B in (1, 2, 4, 32, 64, 256):
        lay = np.zeros(1, T.WITNESS_LAYOUT)
        assert L.bsx_map_witness_layout(C.c_uint32(B), _lib.p(lay)) == T.OK
        assert lay[0].tobytes() == T.map_layout(B).tobytes()
    lay = np.zeros(1, T.WITNESS_LAYOUT)
    assert L.bsx_reduce_witness_layout(_lib.p(lay)) == T.OK
    assert lay[0].tobytes() == T.reduce_layout().tobytes()
    assert L.bsx_map_witness_layout(C.c_uint32(3), _lib.p(lay)) == T.ERR_BAD_ARG
    # the documented production sizes: B = 64 -> 56,096 compact bytes, 449,755 elements per map job
    assert int(T.map_layout(64)["n_bytes"]) == 56096 and int(T.map_layout(64)["n_elements"]) == 449755
    # round 4: COMMIT / SKIP / STEP units (builder.skip / builder.step variables)
    for V in (1, 2, 3, 10, 100, 128, 512):
        assert L.bsx_commit_witness_layout(C.c_uint32(V), _lib.p(lay)) == T.OK
        assert lay[0].tobytes() == T.commit_layout(V).tobytes()
        assert L.bsx_skip_witness_layout(C.c_uint32(V), _lib.p(lay)) == T.OK
        assert lay[0].tobytes() == T.skip_layout(V).tobytes()
        assert L.bsx_header_range_witness_elements(C.c_uint32(32), C.c_uint32(64), C.c_uint32(V)) == T.header_range_witness_elements(32, 64, V)
        assert L.bsx_next_header_witness_elements(C.c_uint32(V)) == T.next_header_witness_elements(V)
    assert L.bsx_step_witness_layout(_lib.p(lay)) == T.OK
    assert lay[0].tobytes() == T.step_layout().tobytes()
    assert L.bsx_commit_witness_layout(C.c_uint32(0), _lib.p(lay)) == T.ERR_BAD_ARG
    assert L.bsx_commit_witness_layout(C.c_uint32(513), _lib.p(lay)) == T.ERR_BAD_ARG
    assert L.bsx_header_range_witness_elements(C.c_uint32(3), C.c_uint32(64), C.c_uint32(100)) == 0
    # sizes quoted in DESIGN.md: the COMMIT unit of a 100-validator commit / of a 512-validator commit
    assert int(T.commit_layout(100)["n_elements"]) == 8 * 48688 + 406 + 958 == 390868 and int(T.commit_layout(512)["n_bytes"]) == 235520


def test_no_gpu_means_no_service():
    L = _lib.lib()
    if L.bsx_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert L.bsx_init(C.c_int(0), C.byref(h)) == T.ERR_NO_DEVICE
    assert b"no CPU fallback" in L.bsx_last_error()
    with pytest.raises(_lib.BsxError):
        _lib.context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "blobstreamx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src and "orc_" not in src, f


def test_witness_manifest_tiles_the_witness_and_matches_the_layout_header():
    """the manifest covers [0, n_elements) exactly once and agrees with include/bsx_layout.h (through its
    Python twin) for every batch size the bins use, and for a reduce node."""
    from blobstreamx_amd.builder import witness_manifest
    for B in (1, 2, 32, 64, 256, 0):
        lay = T.map_layout(B) if B else T.reduce_layout()
        m = witness_manifest(B)
        n = int(lay["n_elements"])
        cover = np.zeros(n, np.int32)
        for e in m:
            for r in range(int(e["repeat"])):
                a = int(e["element_offset"]) + r * int(e["record_stride"])
                cover[a:a + int(e["elements_per_record"])] += 1
            assert e["reference"].decode().split(":")[0] in ("builder.rs", "vars.rs")
        assert (cover == 1).all(), (B, np.nonzero(cover != 1)[0][:5])
        nbits, nw = 8 * int(lay["n_bytes"]), int(lay["n_words"])
        for e in m:
            lo = int(e["element_offset"])
            sec = 0 if lo < nbits else 1 if lo < nbits + nw else 2
            assert sec == int(e["kind"]), e["name"]
        names = [e["name"].decode() for e in m]
        assert len(set(names)) == len(names)
        if B:
            assert names[0] == "ctx.start_header_hash" and "record.data_merkle_root" in names and "data_comm_proof.data_hash_proofs[].leaf" in names
    n = C.c_uint32(0)
    assert _lib.lib().bsx_witness_manifest(C.c_uint32(3), None, C.c_uint32(0), C.byref(n)) == T.ERR_BAD_ARG


def test_unit_manifests_tile_the_commit_skip_and_step_units():
    """Round 4: the manifests of the COMMIT / SKIP / STEP units cover [0, n_elements) exactly once, sit in the
    right sections and name the variables SURVEY P6-P10 / App. B list."""
    from blobstreamx_amd.builder import witness_manifest_section
    cases = [(T.SECTION_COMMIT, V, T.commit_layout(V)) for V in (1, 2, 3, 100, 512)]
    cases += [(T.SECTION_SKIP, V, T.skip_layout(V)) for V in (1, 2, 3, 100, 512)]
    cases += [(T.SECTION_STEP, 0, T.step_layout())]
    for sec, V, lay in cases:
        m = witness_manifest_section(sec, V)
        n = int(lay["n_elements"])
        cover = np.zeros(n, np.int32)
        for e in m:
            for r in range(int(e["repeat"])):
                a = int(e["element_offset"]) + r * int(e["record_stride"])
                cover[a:a + int(e["elements_per_record"])] += 1
        assert (cover == 1).all(), (sec, V, np.nonzero(cover != 1)[0][:5])
        nbits, nw = 8 * int(lay["n_bytes"]), int(lay["n_words"])
        for e in m:
            lo = int(e["element_offset"])
            assert (0 if lo < nbits else 1 if lo < nbits + nw else 2) == int(e["kind"]), e["name"]
        names = [e["name"].decode() for e in m]
        assert len(set(names)) == len(names)
        if sec == T.SECTION_COMMIT:
            for want in ("validator[].pubkey", "validator[].signature (R | s)", "validator[].sha512_digest (SHA512(R|A|M))",
                         "validator[].signature_valid", "validators_hash", "signed_voting_power", "two_thirds_ok (3 * signed > 2 * total)"):
                assert want in names, want
        if sec == T.SECTION_SKIP:
            for want in ("trusted_header_hash (public input)", "data_commitment (public output)", "target.height_proof.proof (4 aunts)",
                         "trusted.validators_hash", "one_third_ok", "trusted.overlap_voting_power (signed the target)"):
                assert want in names, want
    n = C.c_uint32(0)
    assert _lib.lib().bsx_witness_manifest_section(C.c_uint32(T.SECTION_COMMIT), C.c_uint32(0), None, C.c_uint32(0), C.byref(n)) == T.ERR_BAD_ARG
    assert _lib.lib().bsx_witness_manifest_section(C.c_uint32(9), C.c_uint32(1), None, C.c_uint32(0), C.byref(n)) == T.ERR_BAD_ARG


def test_set_tuning_rejects_a_null_context():
    assert _lib.lib().bsx_set_tuning(None, C.c_uint32(T.TUNE_MERKLE_WORKGROUPS), C.c_uint64(512)) == T.ERR_BAD_ARG

