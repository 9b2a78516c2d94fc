"""Time the BASELINE.json parity configurations C3-C5 (model forward, train-mode BN) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd.normalizing_flows.manager import PWQuadManager, PWLinManager
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
cfgs = {
    "C3": ("pwlin", 8, dict(n_pass_through=4, n_cells=4, n_bins=64, NN=[32] * 3, roll_step=4), 1_000_000),
    "C4": ("pwquad", 10, dict(n_cells=5, n_bins=16, NN=[32] * 3), 500_000),
    "C5": ("pwquad", 16, dict(n_cells=8, n_bins=16, NN=[32] * 3), 1_000_000),
    "C5w8": ("pwquad", 16, dict(n_cells=8, n_bins=16, NN=[8] * 3), 1_000_000),
}
for name, (kind, d, kw, N) in cfgs.items():
    if which != "all" and which != name:
        continue
    torch.manual_seed(0)
    NF = (PWQuadManager if kind == "pwquad" else PWLinManager)(n_flow=d)
    NF.create_model(**kw)
    w = torch.rand(N, d, dtype=torch.double, device=dev)
    NF._model.train()
    for mode in ("train", "eval"):
        NF._model.train(mode == "train")
        NF._model.flow_forward(w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            res = NF._model.flow_forward(w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        J = res["planes"][d]
        print(f"{name} {mode}: N={N} {dt*1e3:.2f} ms  {N/dt/1e6:.1f} Msamples/s  meanJ={float(J.mean()):.4f}", flush=True)
