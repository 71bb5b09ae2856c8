"""tools: K = 1, 2 coalesced callers alone (bench.py's latency.concurrent rows).  Round 6 used it for the A/B of round 5's lone-caller
serial short-cut (a second library built with -DBSX_NO_SYNC_FAST_PATH; profiles/r6_lone_caller_ab.txt) before removing the short-cut."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_legs.latency import concurrent_leg
r = concurrent_leg(torch.device("cuda:0"), 32, 64, 100, ks=(1, 2), seconds=0.5, serial=False, forms=())
for x in r["coalesced_shared_context"]:
    print("%s K=%d: %.2f M headers/s  p50 %.3f ms  p99 %.3f ms  requests/set %.1f" % (
        os.path.basename(os.environ.get("BSX_LIB_OVERRIDE", "libbsx.so")), x["threads"], x["headers_per_s"] / 1e6, x["p50_ms"], x["p99_ms"],
        x["requests_per_launch_set"]))
