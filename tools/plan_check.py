"""IntegratePlan (hipGraph replay of a launch set with device-resident generator state) against eager calls, through the
reference API (NF.integrate).     python3 tools/plan_check.py
"""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import integrands                                    # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

dev = torch.device("cuda:0")
f = integrands.Camel()
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
sd0 = copy.deepcopy(NF.best_model.state_dict())

res = {}
for mode in ("graph", "eager"):
    os.environ["NIS_NO_GRAPH"] = "0" if mode == "graph" else "1"
    NF.best_model.load_state_dict(sd0)
    torch.manual_seed(7)
    mom = NF.moments_batches(f, 45, 100000, dev).cpu().numpy()
    sd = {k: v.detach().cpu().numpy().copy() for k, v in NF.best_model.state_dict().items()}
    res[mode] = (mom, sd)
    torch.manual_seed(7)
    print(mode, "integrate:", [float(x) for x in NF.integrate(f, 45, 100000, 0)])
dm = np.max(np.abs(res["graph"][0] - res["eager"][0]) / np.abs(res["eager"][0]))
ds = max(float(np.max(np.abs(res["graph"][1][k] - res["eager"][1][k]) / (np.abs(res["eager"][1][k]) + 1e-9))) for k in sd0)
print(f"graph vs eager: moments max rel diff {dm:.2e} (45 iterations = 2 replays of 20 + 5 eager), state_dict {ds:.2e}")
print("num_batches_tracked:", {k: int(v) for k, v in res["graph"][1].items() if k.endswith("0.num_batches_tracked")})

for mode in ("graph", "eager"):
    os.environ["NIS_NO_GRAPH"] = "0" if mode == "graph" else "1"
    for nitn in (20, 20, 100, 1000):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = NF.integrate(f, nitn, 1_000_000, 0)
        dt = time.perf_counter() - t0
        print(f"NF.integrate(camel, {nitn}, 1e6) {mode}: {dt * 1e3:8.2f} ms = {dt / nitn * 1e6:6.2f} us/iteration = {nitn * 1e6 / dt:.3e} samples/s  -> {float(r[0]):.6f} +- {float(r[1]):.2e}", flush=True)
