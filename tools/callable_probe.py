"""integrate() with a plain Python callable (the reference's README form) against the built-in integrand.  python tools/callable_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWQuadManager


def camel(x):
    return torch.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / (0.2 ** 2)) + \
        torch.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / (0.2 ** 2))


torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
for name, f in (("built-in Camel()", integrands.Camel()), ("python callable", camel)):
    for _ in range(2):
        NF.integrate(f, 10, 1_000_000, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sig, err = NF.integrate(f, 10, 1_000_000, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:18s}: integrate(f, 10, 1e6) {dt * 1e3:8.3f} ms  ({10e6 / dt / 1e9:6.2f} G samples/s)  {float(sig):.6f} +- {float(err):.2e}", flush=True)
