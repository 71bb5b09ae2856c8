"""tools: bench.py's range sweep alone (compact form; `w` as argv[1] for the witness form)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import bench
dev = torch.device("cuda:0")
wit = len(sys.argv) > 1 and sys.argv[1] == "w"
r = bench.range_sweep_leg(dev, 32, 64, 100, rs=(1, 4, 16, 64) if wit else (1, 4, 16, 64, 128, 256), witness=wit)
for row in r["by_ranges"]:
    print("R=%4d  %8.2f M headers/s  %.3f ms/step" % (row["ranges"], row["headers_per_s"] / 1e6, row["ms_per_step"]))
