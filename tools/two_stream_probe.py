"""Does overlapping two independent launch sets on two streams pay?  Two copies of the C2 model (own workspaces) alternate on
two HIP streams, so one set's HBM-bound statistics passes can run beside the other's fp64-issue-bound apply kernels.

    python3 tools/two_stream_probe.py [G]
"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NIS_NO_MEGA"] = "1"
from nf_amd import integrands                                    # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 1_000_000
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
f = integrands.Camel()
models = [NF.best_model, copy.deepcopy(NF.best_model)]
for m in models:
    m.train()
engs = [m.engine() for m in models]
toks = [e.pack_token() for e in engs]
ws = [torch.rand(G, N, 2, dtype=torch.float32, device=dev) for _ in range(2)]
outs = [torch.zeros((G, 4), dtype=torch.double, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(n_sets, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_sets):
        i = k & 1
        if two:
            with torch.cuda.stream(streams[i]):
                engs[i].integrate_steps(ws[i], True, f.nis_kind, f.nis_params, outs[i], token=toks[i])
        else:
            engs[i].integrate_steps(ws[i], True, f.nis_kind, f.nis_params, outs[i], token=toks[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for n_sets in (2, 4, 40, 200):
    for two in (False, True, False, True):
        us = run(n_sets, two)
        print(f"{'two streams' if two else 'one stream '}  {n_sets:3d} sets of {G}: {us / n_sets / G:7.2f} us/step = {N * G * n_sets / us * 1e6:.3e} samples/s", flush=True)
