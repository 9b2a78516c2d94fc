"""Persistent whole-step kernel (flow_mega.hip) against the launch-per-phase path and the oracle, plus timings.

    python3 tools/mega_check.py [N] [G]
"""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import integrands                                    # noqa: E402
from nf_amd import _lib as _nis_lib                               # noqa: E402
assert _nis_lib.lib().nis_diag_build() == 1, "this probe exercises a quarantined variant: run it with NIS_LIB=nf_amd/lib/libnis_hip_diag.so (python -m nf_amd.build --diag)"
from nf_amd.flow import philox_rows                              # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
f = integrands.Camel()


def fresh():
    torch.manual_seed(0)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    m = NF.best_model
    m.train()
    return NF, m


def run(model, w, mega):
    os.environ["NIS_NO_MEGA"] = "0" if mega else "1"
    eng = model.engine()
    out = torch.zeros((w.shape[0], 4), dtype=torch.double, device=dev)
    eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    torch.cuda.synchronize()
    eng.resident_status(dev)
    return out.cpu().numpy(), {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


NF, base = fresh()
w = torch.rand(G, N, 2, dtype=torch.float32, device=dev)
ma, sa = run(copy.deepcopy(base), w, True)
mb, sb = run(copy.deepcopy(base), w, False)
print("moments mega vs launches: max rel diff", np.max(np.abs(ma - mb) / np.maximum(np.abs(mb), 1e-300)))
print(" mega    ", ma[0])
print(" launches", mb[0])
worst = max((float(np.max(np.abs(sa[k] - sb[k]) / np.maximum(np.abs(sb[k]), 1e-12))), k) for k in sa)
print("state_dict (running statistics) max rel diff", worst, sa[worst[1]], sb[worst[1]])
ma2, _ = run(copy.deepcopy(base), w, True)
print("bitwise reproducible:", bool(np.array_equal(ma, ma2)))

# oracle at a size it finishes quickly
if N <= 200000:
    from oracle import flow as oflow
    params = {k: v.detach().cpu().numpy() for k, v in base.state_dict().items()}
    steps = oflow.plan_pwquad(2, 2, 4)
    res = oflow.integrate(oflow.camel, params, steps, [w[g].cpu().numpy() for g in range(G)])
    print("oracle integrate:", res[0], res[1])
    var = ma[:, 2] / (ma[:, 0] - 1)
    print("mega   integrate:", np.sum(ma[:, 1] / var) / np.sum(1 / var), np.sqrt(1 / np.sum(1 / var)) / np.sqrt(N * G))

# in-kernel draws: the draws that were used == philox_rows, and feeding them back reproduces the records bit for bit
model = copy.deepcopy(base)
eng = model.engine()
os.environ["NIS_NO_MEGA"] = "0"
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
wout = torch.zeros((G, N, 2), dtype=torch.float32, device=dev)
eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, seed=1234, stream0=77, w_out=wout)
torch.cuda.synchronize()
eng.resident_status(dev)
pr = philox_rows(1234, 77, G, N, 2, dev)
print("draws == philox_rows:", bool(torch.equal(wout, pr)), " mean", float(pr.mean()), " min", float(pr.min()), " max", float(pr.max()))
mc, _ = run(copy.deepcopy(base), pr, True)
print("rng records == fed records (bitwise):", bool(np.array_equal(out.cpu().numpy(), mc)))
var = mc[:, 2] / (mc[:, 0] - 1)
print("rng integral", np.sum(mc[:, 1] / var) / np.sum(1 / var), "truth 0.2323236")


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


model = copy.deepcopy(base)
eng = model.engine()
tok = eng.pack_token()
for name, fn in [("launches fed", lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
                 ("mega fed", lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
                 ("mega rng", lambda: eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok))]:
    os.environ["NIS_NO_MEGA"] = "1" if name.startswith("launches") else "0"
    for n in (1, 3, 30, 100):
        wall, per = timeit(fn, n)
        print(f"{name:13s} {n:3d} sets of {G}: wall {wall / n / G:7.2f} us/step = {N * G * n / wall * 1e6:.3e} samples/s | per set us: "
              + " ".join(f"{p:6.0f}" for p in per[:6]) + (f" ... last {per[-1]:.0f}" if n > 6 else ""), flush=True)
eng.resident_status(dev)
