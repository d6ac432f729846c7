"""bench.py's run_host_path on its own (is its batch-of-8 figure a property of that function or of what ran before it?)"""
import argparse, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
args = argparse.Namespace(ny=2400, nx=3600)
for k in range(2):
    rec = bench.run_host_path(torch.device("cuda", 0), args)
    print(k, {q: round(v, 3) for q, v in rec.items() if isinstance(v, float)}, flush=True)
