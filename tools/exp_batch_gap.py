import json, os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
r = bench.concurrent_leg(dev, 32, 64, 100, ks=(1, 16), seconds=0.4, serial=False)
print("gap", os.environ.get("BSX_BATCH_GAP_NS"), [(x["threads"], round(x["headers_per_s"] / 1e6, 1), round(x["p50_ms"], 3), round(x["p99_ms"], 3), round(x["requests_per_launch_set"], 1)) for x in r["coalesced_shared_context"]], flush=True)
h = bench.hint_concurrent_leg(dev, 32, 64, 100, reps=60)
print("   hint", {k: (round(v["median_ms"], 3), round(v["min_ms"], 3), round(v["p90_ms"], 3)) for k, v in h["coalesced"].items() if isinstance(v, dict) and "median_ms" in v}, h["coalesced"]["requests_per_launch_set"], flush=True)
