"""Launch-per-phase path: one chain vs several chains on side streams vs in-kernel draws vs the opt-in persistent kernel.
Parity between the variants (rtol 1e-12 on the moment records) + timings.

    python3 tools/streams_check.py [N] [G]
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
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
base = NF.best_model
base.train()


def setenv(streams, mega):
    os.environ["NIS_STREAMS"] = str(streams)
    os.environ["NIS_MEGA"] = "1" if mega else "0"


def run(w, streams, mega, rng=None):
    setenv(streams, mega)
    model = copy.deepcopy(base)
    eng = model.engine()
    out = torch.zeros((G, 4), dtype=torch.double, device=dev)
    wout = None
    if rng is not None:
        wout = torch.zeros((G, N, 2), dtype=torch.float32, device=dev)
        eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, seed=rng[0], stream0=rng[1], w_out=wout)
    else:
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    torch.cuda.synchronize()
    eng.resident_status(dev)
    return out.cpu().numpy(), {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}, wout


def cmp(name, a, b):
    ma, sa, _ = a
    mb, sb, _ = b
    dm = np.max(np.abs(ma - mb) / np.maximum(np.abs(mb), 1e-300))
    ds = max(float(np.max(np.abs(sa[k] - sb[k]) / (np.abs(sb[k]) + 1e-9))) for k in sa)
    print(f"{name}: moments max rel diff {dm:.2e}, state_dict max diff {ds:.2e}", "OK" if dm < 1e-12 and ds < 1e-9 else "MISMATCH")


w = torch.rand(G, N, 2, dtype=torch.float32, device=dev)
one = run(w, 1, False)
two = run(w, 2, False)
thr = run(w, 3, False)
cmp("2 chains vs 1", two, one)
cmp("3 chains vs 1", thr, one)
print("2 chains bitwise reproducible:", bool(np.array_equal(two[0], run(w, 2, False)[0])))
r1 = run(None, 2, False, rng=(1234, 77))
pr = philox_rows(1234, 77, G, N, 2, dev)
print("launch-path draws == philox_rows:", bool(torch.equal(r1[2], pr)))
cmp("in-kernel draws vs fed draws", r1, run(pr, 1, False))
mg = run(pr, 1, True)
cmp("persistent kernel vs launches", mg, run(pr, 1, False))
var = r1[0][:, 2] / (r1[0][:, 0] - 1)
print("integral (in-kernel draws)", np.sum(r1[0][:, 1] / var) / np.sum(1 / var), "truth 0.2323236")


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


model = copy.deepcopy(base)
eng = model.engine()
tok = eng.pack_token()
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
for name, streams, mega, fn in [
        ("fed, 1 chain", 1, False, lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
        ("fed, 2 chains", 2, False, lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
        ("fed, 3 chains", 3, False, lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
        ("fed, 4 chains", 4, False, lambda: eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)),
        ("rng, 1 chain", 1, False, lambda: eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok)),
        ("rng, 2 chains", 2, False, lambda: eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok)),
        ("rng, 3 chains", 3, False, lambda: eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok)),
        ("rng, persistent", 1, True, lambda: eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok))]:
    setenv(streams, mega)
    res = []
    for n in (1, 2, 20, 100):
        us = timeit(fn, n)
        res.append(f"{n}: {us / n / G:6.2f}")
    print(f"{name:16s} us/step by number of sets of {G}:  " + "   ".join(res) + f"   -> {N / (us / 100 / G) * 1e6:.3e} samples/s", flush=True)
