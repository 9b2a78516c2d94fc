"""A few training minibatches of bench.py's train_step leg (profiling target):  python3 tools/run_train_leg.py C2|C5 [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
r = bench.bench_train_step(name, 0, 1, torch.device("cuda:0"), steps=steps)
print({k: r[k] for k in ("value", "ms_per_step", "loss")})
