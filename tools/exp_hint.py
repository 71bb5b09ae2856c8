import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.hint_concurrent_leg(dev, 32, 64, 100), indent=1))
