"""tools: bench.py's concurrent leg (K threads x own context x bsx_header_range) and the single-call latency alone."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import bench
dev = torch.device("cuda:0")
from blobstreamx_amd import _lib
_lib.lib()
l = bench.latency_leg(dev, 32, 64, 100)
print("latency median %.3f min %.3f" % (l["output_only_ms"]["median"], l["output_only_ms"]["min"]))
c = bench.concurrent_leg(dev, 32, 64, 100)
print([(r["threads"], round(r["headers_per_s"] / 1e6, 1), round(r["p50_ms"], 2)) for r in c["by_threads"]])
