#!/usr/bin/env python3
"""K native threads x bsx_header_range on one coalescing context for a short while, for a rocprofv3 trace
(rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -o ct -- python tools/concurrent_trace.py [K=16])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = bench.concurrent_leg(torch.device("cuda:0"), 32, 64, 100, ks=(K,), seconds=0.15, serial=False, forms=())
x = r["coalesced_shared_context"][0]
print(K, round(x["headers_per_s"] / 1e6, 1), x["p50_ms"], x["requests_per_launch_set"], x["worker_us_per_set"])
