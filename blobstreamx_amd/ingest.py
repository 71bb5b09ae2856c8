"""Wire-format ingest: Tendermint RPC JSON -> packed layouts, through the C ABI (bsx_ingest_*).

Mirrors what the reference's fetcher decodes with serde + tendermint-rs (circuits/input.rs:19-27,67-145;
circuits/fetcher.rs:44-58,89-132) and its fixture mode (`InputDataMode::Fixture`, input.rs:97-101): the files under
circuits/fixtures/mocha-4 are accepted verbatim.  Host-side byte formatting only — usable without a GPU."""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import types as T


def _check(rc):
    if rc != T.OK:
        raise _lib.BsxError(rc, _lib.lib().bsx_ingest_last_error().decode(errors="replace"))


def header_from_json(text):
    raw = text.encode() if isinstance(text, str) else bytes(text)
    h = np.zeros(1, T.HEADER)
    height = C.c_uint64(0)
    _check(_lib.lib().bsx_ingest_header_json(raw, C.c_size_t(len(raw)), _lib.p(h), C.byref(height)))
    return h[0], height.value


def signed_block_from_json(text, v_max, validators_text=None):
    """-> dict(header, height, block_hash, validators[v_max], n_validators)"""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    vraw = None if validators_text is None else (validators_text.encode() if isinstance(validators_text, str) else bytes(validators_text))
    h = np.zeros(1, T.HEADER)
    bh = np.zeros(32, np.uint8)
    vals = np.zeros(v_max, T.VALIDATOR)
    n = C.c_uint32(0)
    height = C.c_uint64(0)
    _check(_lib.lib().bsx_ingest_signed_block_json(raw, C.c_size_t(len(raw)), vraw, C.c_size_t(len(vraw) if vraw else 0), _lib.p(h),
                                                   _lib.p(bh), _lib.p(vals), C.c_uint32(v_max), C.byref(n), C.byref(height)))
    return dict(header=h[0], height=height.value, block_hash=bh.tobytes(), validators=vals, n_validators=n.value)


def data_commitment_from_json(text):
    raw = text.encode() if isinstance(text, str) else bytes(text)
    out = np.zeros(32, np.uint8)
    _check(_lib.lib().bsx_ingest_data_commitment_json(raw, C.c_size_t(len(raw)), _lib.p(out)))
    return out.tobytes()


class FixtureFetcher:
    """The reference's fixture mode (tendermintx InputDataFetcher with InputDataMode::Fixture, used at
    circuits/input.rs:97-101 and by the builder tests, builder.rs:462-485): `<dir>/<height>/signed_block.json` and
    `<dir>/<start>-<end>/data_commitment.json`."""

    def __init__(self, fixture_path, v_max=4):
        self.path, self.v_max = fixture_path, v_max

    def signed_block(self, height):
        with open(os.path.join(self.path, str(height), "signed_block.json"), "rb") as f:
            return signed_block_from_json(f.read(), self.v_max)

    def get_data_commitment(self, start_block, end_block):
        # circuits/input.rs:67-72: a dummy (zero) commitment when the range is empty
        if end_block <= start_block:
            return bytes(32)
        with open(os.path.join(self.path, f"{start_block}-{end_block}", "data_commitment.json"), "rb") as f:
            return data_commitment_from_json(f.read())

    def headers(self, first, last):
        return np.array([self.signed_block(h)["header"] for h in range(first, last + 1)], dtype=T.HEADER)
