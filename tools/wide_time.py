"""Train-mode forward time of the BASELINE wide configs (C3 / C4 / C5 at their per-GPU batch), events around 5 forwards.
    python3 tools/wide_time.py [names]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["C3", "C4", "C5"]):
    kind, d, kw, N, _, _ = bench.CONFIGS[name]
    torch.manual_seed(10)
    NF = (bench.PWQuadManager if kind == "pwquad" else bench.PWLinManager)(n_flow=d)
    NF.create_model(**kw)
    m = NF.best_model
    m.train()
    w = torch.rand(N, d, dtype=torch.float32, device=dev)
    for _ in range(2):
        res = m.flow_forward(w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        res = m.flow_forward(w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: N={N} train forward {ms:8.3f} ms  {N / ms / 1e3:8.2f} Msamples/s  <J>={float(res['planes'][d].mean()):.6f}", flush=True)
