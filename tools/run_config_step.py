"""A few iterations of integrate()'s loop body for a BASELINE config (bench.py's `configs` leg), for rocprofv3 passes:
    python3 tools/run_config_step.py C5 [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
kind, d, kw, N, _, _ = bench.CONFIGS[name]
torch.manual_seed(10)
NF = (bench.PWQuadManager if kind == "pwquad" else bench.PWLinManager)(n_flow=d)
NF.create_model(**kw)
NF.best_model.train()
f, _ = bench.config_integrand(name)
mom = NF.moments_batches(f, iters, N, dev)
torch.cuda.synchronize()
print(mom.cpu().numpy())
