"""bench.py's keyset_churn leg alone: python tools/exp_churn.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.keyset_churn_leg(torch.device("cuda:0"), 32, 64, 100), indent=1))
