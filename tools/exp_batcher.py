"""Run bench.py's coalescing legs alone (latency.concurrent, hint_concurrent) and print them: python tools/exp_batcher.py [sweep]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

J, B, V = 32, 64, 100
dev = torch.device("cuda:0")
out = {}
if "sweep2" in sys.argv:          # lanes x window (round 5, second pass)
    for lanes in (2, 3, 4, 6):
        for win in (20, 50, 150):
            r = bench.concurrent_leg(dev, J, B, V, ks=(16, 64), seconds=0.4, serial=False, window_us=win, n_lanes=lanes)
            out["l%d_w%d" % (lanes, win)] = [(x["threads"], round(x["headers_per_s"] / 1e6, 1), round(x["p50_ms"], 3), round(x["p99_ms"], 3), round(x["requests_per_launch_set"], 1))
                                            for x in r["coalesced_shared_context"]]
            print("l%d_w%d" % (lanes, win), out["l%d_w%d" % (lanes, win)], flush=True)
elif "cap" in sys.argv:           # round 6: does capping the set size break the callers' lockstep?  (K callers, max_requests per set, lanes)
    for K, caps, lanes in ((16, (0, 4, 6, 8), (3, 4)), (64, (0, 16, 22), (3, 4))):
        for ln in lanes:
            for mr in caps:
                r = bench.concurrent_leg(dev, J, B, V, ks=(K,), seconds=0.4, serial=False, forms=(), max_requests=mr, n_lanes=ln)
                x = r["coalesced_shared_context"][0]
                print("K=%d lanes=%d max_requests=%d: %.1f M headers/s  p50 %.3f  p99 %.3f  requests/set %.1f  %s" % (
                    K, ln, mr, x["headers_per_s"] / 1e6, x["p50_ms"], x["p99_ms"], x["requests_per_launch_set"], x["worker_us_per_set"]), flush=True)
elif "sweep" in sys.argv:
    for name, kw in (("w50_l3", {}), ("w100_l2", dict(window_us=100, n_lanes=2)), ("w50_l3_pinned", dict(pinned=True))):
        r = bench.concurrent_leg(dev, J, B, V, ks=(1, 16, 64), seconds=0.4, serial=False, **kw)
        out[name] = [(x["threads"], round(x["headers_per_s"] / 1e6, 1), round(x["p50_ms"], 3), round(x["p99_ms"], 3), round(x["requests_per_launch_set"], 1), x["worker_us_per_set"])
                     for x in r["coalesced_shared_context"]]
    print(json.dumps(out, indent=1))
else:
    out = {"concurrent": bench.concurrent_leg(dev, J, B, V), "hint_concurrent": bench.hint_concurrent_leg(dev, J, B, V)}
    print(json.dumps(out, indent=1))
