#!/usr/bin/env python3
"""The fused / materialised Poseidon witness-commitment leg of bench.py alone (R ranges of header_range_2048)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from blobstreamx_amd import _lib
_lib.lib()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
d = bench.commitment_leg(dev, 32, 64, 100, bench.calibrate(dev), R=R)
print(json.dumps({k: d[k] for k in ("fused_ms", "materialised_ms", "headers_per_s_fused", "permutations", "pipeline_caps_mode")}), d["roofline"]["achieved"], d["roofline"]["frac"])
