"""Is the first launch set after a synchronize slow because the host cannot feed the first short kernels fast enough?
Capture one launch set into a hipGraph (torch.cuda.CUDAGraph) and compare per-set times with eager launches.

    python3 tools/graph_probe.py [G]
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
f = integrands.Camel()
m = NF.best_model
m.train()
eng = m.engine()
tok = eng.pack_token()
w = torch.rand(G, N, 2, dtype=torch.float32, device=dev)
out = torch.zeros((G, 4), dtype=torch.double, device=dev)


def eager():
    eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)


for _ in range(3):
    eager()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    eager()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        eager()
torch.cuda.synchronize()
ref = out.clone()
out.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay reproduces the eager records:", bool(torch.equal(ref, out)))


def timeit(fn, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(n):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    return wall, [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]


for name, fn in (("eager", eager), ("graph", g.replay), ("eager", eager), ("graph", g.replay)):
    for idle, n in ((0.0, 2), (0.0, 2), (0.5, 2), (0.0, 20)):
        time.sleep(idle)
        wall, per = timeit(fn, n)
        print(f"{name} idle {idle:.1f}s {n:2d} sets of {G}: wall {wall:7.0f} us = {wall / n / G:6.2f} us/step | per set: " + " ".join(f"{p:5.0f}" for p in per[:5]), flush=True)
