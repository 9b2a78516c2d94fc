"""User-level `integrate(f, nitn, neval)` wall time (draws included) for the C2 model, nitn in {10, 100}."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWQuadManager
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
f = integrands.Camel()
for nitn in (10, 100):
    NF.integrate(f, nitn, 1_000_000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sig, err = NF.integrate(f, nitn, 1_000_000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"integrate(camel, {nitn}, 1e6): {dt * 1e3:.3f} ms  {nitn * 1e6 / dt / 1e9:.2f} Gsamples/s  -> {float(sig):.6f} +- {float(err):.2e}", flush=True)
lam = lambda x: torch.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / 0.04) + torch.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / 0.04)
NF.integrate(lam, 10, 1_000_000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    sig, err = NF.integrate(lam, 10, 1_000_000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"integrate(python callable, 10, 1e6): {dt * 1e3:.3f} ms  {1e7 / dt / 1e9:.2f} Gsamples/s  -> {float(sig):.6f} +- {float(err):.2e}", flush=True)
