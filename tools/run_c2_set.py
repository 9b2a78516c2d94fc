"""A few EAGER launch sets of the C2 step with in-kernel draws (the bench's workload without the graph replay), for rocprofv3
passes:   python3 tools/run_c2_set.py [G] [sets]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import integrands                                    # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SETS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
m = NF.best_model
m.train()
eng = m.engine()
f = integrands.Camel()
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
tok = eng.pack_token()
for k in range(SETS):
    eng.integrate_steps_rng(G, 1_000_000, dev, f.nis_kind, f.nis_params, out, 1, k * G, token=tok)
torch.cuda.synchronize()
print(out[:2].cpu().numpy())
