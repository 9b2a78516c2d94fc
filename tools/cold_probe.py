"""How does the C2 launch set behave right after an idle period?  Per-set GPU time (events on the launch stream) of n
consecutive launch sets of G steps, after sleeping `idle` seconds -- the driver's `bench.py --steps 20 --warmup 5` times
two sets right after a one-set warm-up on a cold GPU.

    python3 tools/cold_probe.py [G]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import integrands                                    # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 1_000_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
model = NF.best_model
model.train()
eng = model.engine()
f = integrands.Camel()
w = torch.rand(G, N, 2, dtype=torch.float32, device=dev)
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
tok = eng.pack_token()


def run_sets(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(n):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
        ev[k + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
    return wall * 1e6, per


for idle, n in [(0.0, 1), (0.0, 3), (1.0, 3), (1.0, 3), (0.05, 3), (1.0, 30), (0.0, 100)]:
    time.sleep(idle)
    wall, per = run_sets(n)
    head = " ".join(f"{p:6.0f}" for p in per[:12])
    print(f"idle {idle:4.2f}s  {n:3d} sets of {G}: wall {wall:8.0f} us = {wall / n / G:6.1f} us/step = {N * G * n / wall * 1e6:.3e} samples/s | per set us: {head}"
          + (f" ... last {per[-1]:.0f}" if n > 12 else ""), flush=True)
