"""pytest configuration: the `gpu` marker + shared fixtures.

`-m "not gpu"` runs here without a GPU: oracle vs golden vectors, host logic, C-ABI symbol check.
`-m gpu` runs on an MI355X: HIP kernels vs the oracle, through the C ABI (libbsx.so).
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from blobstreamx_amd import types as T  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "mocha4.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def mocha(golden):
    """The five fixture blocks 10000..10004 as packed headers + hashes + commits."""
    heights = list(range(10000, 10005))
    headers = np.array([T.pack_header([bytes.fromhex(f) for f in golden["blocks"][str(h)]["fields"]]) for h in heights])
    hashes = [bytes.fromhex(golden["blocks"][str(h)]["header_hash"]) for h in heights]
    commits = []
    for h in heights:
        b = golden["blocks"][str(h)]
        vals = np.zeros(4, T.VALIDATOR)
        for i, v in enumerate(b["validators"]):
            vals[i]["pubkey"] = np.frombuffer(bytes.fromhex(v["pubkey"]), np.uint8)
            vals[i]["voting_power"] = v["power"]
            vals[i]["enabled"] = 1
            vals[i]["present_on_trusted"] = 1
        for s in b["commit"]["signatures"]:
            i, m = s["validator_index"], bytes.fromhex(s["sign_bytes"])
            vals[i]["signature"] = np.frombuffer(bytes.fromhex(s["signature"]), np.uint8)
            vals[i]["message"][:len(m)] = np.frombuffer(m, np.uint8)
            vals[i]["message_len"] = len(m)
            vals[i]["is_signed"] = 1
        commits.append(vals)
    return dict(heights=heights, headers=headers, hashes=hashes, commits=commits, first_height=10000, latest=10006)
