"""ms per integrate() iteration of the wide BASELINE configs (fused Normal integrand, in-kernel draws, G = 3 per launch set), for A/B
runs of library variants:   [NIS_LIB=...] [NIS_NO_REGBIJ=1] python3 tools/wide_ab.py [C3 C4 C5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
names = sys.argv[1:] or ["C3", "C4", "C5"]
tag = os.path.basename(os.environ.get("NIS_LIB", "default")) + (" NO_REGBIJ" if os.environ.get("NIS_NO_REGBIJ") else "")
for name in names:
    kind, d, kw, N, _, _ = bench.CONFIGS[name]
    torch.manual_seed(10)
    NF = (bench.PWQuadManager if kind == "pwquad" else bench.PWLinManager)(n_flow=d)
    NF.create_model(**kw)
    m = NF.best_model
    m.train()
    eng = m.engine()
    f = bench.integrands.Normal(0.5, 0.1, d)
    G = 3
    out = torch.zeros((G, 4), dtype=torch.double, device=dev)
    for k in range(2):
        eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, k * G)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for k in range(reps):
        eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, (k + 2) * G)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps / G
    print(f"[{tag}] {name}: {ms:8.3f} ms per iteration  ({N / ms / 1e3:.3e} samples/s)   mean f*J {float(out[0, 1]):.6e}", flush=True)
