"""tools: K = 1, 2 coalesced callers alone (bench.py's latency.concurrent rows) — the A/B of the lone-caller serial path of round 5
(BSX_LIB_OVERRIDE=.../libbsx_nofast.so = built with -DBSX_NO_SYNC_FAST_PATH: every synchronous call goes through the batcher)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_legs.latency import concurrent_leg
r = concurrent_leg(torch.device("cuda:0"), 32, 64, 100, ks=(1, 2), seconds=0.5, serial=False, forms=())
for x in r["coalesced_shared_context"]:
    print("%s K=%d: %.2f M headers/s  p50 %.3f ms  p99 %.3f ms  requests/set %.1f" % (
        "no-fast-path build" if "nofast" in os.environ.get("BSX_LIB_OVERRIDE", "") else "product", x["threads"], x["headers_per_s"] / 1e6, x["p50_ms"], x["p99_ms"],
        x["requests_per_launch_set"]))
