"""CPU suite: packed wire headers (include/bsx.h bsx_pack_headers / bsx_unpack_headers, csrc/wire.cpp) — host code of the product
library, no GPU: the fixture headers and synthetic ones round-trip; inconsistent blocks are refused."""
import ctypes as C

import numpy as np
import pytest

import synth
from blobstreamx_amd import _lib
from blobstreamx_amd import batcher as BT
from blobstreamx_amd import types as T


def test_fixture_headers_round_trip_and_shrink(mocha):
    hdr = mocha["headers"]
    pk = BT.pack_headers(hdr)
    n, nb = np.frombuffer(pk[:8].tobytes(), np.uint32)
    assert n == 5 and nb == pk.size
    # 394 bytes of fields + 14 length bytes per header, + 4 bytes of offset, + the block's 8-byte head and padding
    off = np.frombuffer(pk[8:8 + 4 * 6].tobytes(), np.uint32)
    assert list(np.diff(off)) == [int(h["len"].sum()) + 14 for h in hdr] and all(380 <= d <= 420 for d in np.diff(off))
    back = BT.unpack_headers(pk)
    assert back.tobytes() == np.ascontiguousarray(hdr).tobytes()


def test_synthetic_headers_round_trip_incl_empty_and_overlong_fields():
    w = synth.Workload(3, 1, 4, 16, v=3)
    hdr = np.ascontiguousarray(w.headers[0]).copy()
    hdr[3]["len"][5] = 0                     # an empty field (hashed as the empty leaf)
    hdr[3]["hash"][0][:] = 0
    pk = BT.pack_headers(hdr)
    assert pk.size < 0.85 * hdr.nbytes
    assert BT.unpack_headers(pk).tobytes() == hdr.tobytes()
    # an over-long length byte travels as it is (the request reports BSX_ERR_BAD_HEADER on the device, like a 512-byte record);
    # the field's bytes stop at its capacity on both sides
    hdr[7]["len"][3] = 60
    back = BT.unpack_headers(BT.pack_headers(hdr))
    assert back[7]["len"][3] == 60 and back[7]["time"].tobytes() == hdr[7]["time"].tobytes()
    assert back[8].tobytes() == hdr[8].tobytes()


def test_inconsistent_blocks_are_refused():
    w = synth.Workload(3, 1, 2, 8, v=3)
    pk = BT.pack_headers(w.headers[0])
    L = _lib.lib()
    n = C.c_uint64(0)

    def check(blk):
        blk = np.ascontiguousarray(blk, np.uint8)
        return L.bsx_unpack_headers(_lib.p(blk), C.c_uint64(blk.size), None, C.c_uint64(0), C.byref(n))
    assert check(pk) == T.OK and n.value == w.headers[0].size
    bad = pk.copy(); bad[8 + 4 * 2:8 + 4 * 2 + 4] = np.frombuffer((7).to_bytes(4, "little"), np.uint8)      # offsets run backwards
    assert check(bad) == T.ERR_BAD_HEADER
    bad = pk.copy(); bad[4:8] = np.frombuffer((pk.size + 64).to_bytes(4, "little"), np.uint8)               # claims more bytes than there are
    assert check(bad) == T.ERR_BAD_HEADER
    bad = pk.copy(); bad[0:4] = 0                                                                         # no headers
    assert check(bad) == T.ERR_BAD_HEADER
    assert check(pk[:100]) == T.ERR_BAD_HEADER and check(pk[:8]) == T.ERR_BAD_ARG
    out = np.zeros(3, T.HEADER)                                                                           # too small an output
    assert L.bsx_unpack_headers(_lib.p(pk), C.c_uint64(pk.size), _lib.p(out), C.c_uint64(3), C.byref(n)) == T.ERR_BAD_ARG
    with pytest.raises(_lib.BsxError):
        BT.unpack_headers(pk[:100])
