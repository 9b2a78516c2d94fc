"""Is the FIRST replay of a captured launch set slower than later ones?  (bench.py --steps 20 times exactly one replay.)
    python3 tools/first_replay_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nf_amd import integrands
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = bench.PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3, dev=0)
m = NF.best_model; m.train(); eng = m.engine(); f = integrands.Camel()
G, N = 20, 1_000_000
plan = eng.plan(G, N, dev, f.nis_kind, f.nis_params)
plan.reset(1234, 0)
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
scratch = torch.zeros((G, 4), dtype=torch.double, device=dev)
eng.integrate_steps_rng(5, N, dev, f.nis_kind, f.nis_params, scratch[:5], 1234, 1 << 40, token=plan.token)   # the driver's 5 warm-up steps
torch.cuda.synchronize()
for k in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run(out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"replay {k}: {dt * 1e6:8.1f} us  -> {G * N / dt:.3e} samples/s", flush=True)
    time.sleep(0.05 if k % 2 else 0.0)
# eager set for comparison, idle stream
for k in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1234, 100 * k, token=plan.token)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"eager  {k}: {dt * 1e6:8.1f} us  -> {G * N / dt:.3e} samples/s", flush=True)
