"""Driver-invocation rehearsal: 20 steps as launch sets replayed from hipGraphs on two free-running lanes (two model copies,
own workspaces), against one lane / eager.      python3 tools/lanes_probe.py [G]
"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NIS_STREAMS"] = "1"
from nf_amd import integrands                                    # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10
STEPS = 20
N = 1_000_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
f = integrands.Camel()
models = [NF.best_model, copy.deepcopy(NF.best_model)]
lanes = []
for m in models:
    m.train()
    eng = m.engine()
    tok = eng.pack_token()
    out = torch.zeros((G, 4), dtype=torch.double, device=dev)
    st = torch.cuda.Stream()

    def call(eng=eng, out=out, tok=tok):
        eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0, token=tok)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        call()
        call()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            call()
    torch.cuda.synchronize()
    lanes.append((st, g, call))


def run(n_sets, n_lanes, graph):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_sets):
        st, g, call = lanes[k % n_lanes]
        with torch.cuda.stream(st):
            if graph:
                g.replay()
            else:
                call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


n_sets = STEPS // G
for rep in range(3):
    for n_lanes in (1, 2):
        for graph in (False, True):
            time.sleep(0.2)
            us = run(n_sets, n_lanes, graph)
            print(f"{STEPS} steps as {n_sets} sets of {G}, {n_lanes} lane(s), {'graph' if graph else 'eager'}: {us:7.0f} us = {us / STEPS:6.2f} us/step = {N * STEPS / us * 1e6:.3e} samples/s", flush=True)
us = run(200 // G * 5, 2, True)
print(f"sustained, 2 lanes graph: {us / (200 // G * 5 * G):6.2f} us/step")
us = run(200 // G * 5, 1, True)
print(f"sustained, 1 lane graph: {us / (200 // G * 5 * G):6.2f} us/step")
