"""integrate()'s loop body for a wide BASELINE config, fused rider, by iterations per launch set.  python3 tools/wide_steps_time.py C3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
kind, d, kw, N, _, _ = bench.CONFIGS[name]
torch.manual_seed(10)
NF = (bench.PWQuadManager if kind == "pwquad" else bench.PWLinManager)(n_flow=d)
NF.create_model(**kw)
m = NF.best_model
m.train()
eng = m.engine()
f = bench.integrands.Normal(0.5, 0.1, d)
for G in (1, 2, 3, 6):
    w = torch.rand(G, N, d, dtype=torch.float32, device=dev)
    out = torch.zeros((G, 4), dtype=torch.double, device=dev)
    for _ in range(2):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, 6 // G)
    e0.record()
    for _ in range(reps):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} fused G={G}: {e0.elapsed_time(e1) / reps / G:8.3f} ms per iteration", flush=True)
w = torch.rand(N, d, dtype=torch.float32, device=dev)
for _ in range(2):
    m.flow_forward(w)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    m.flow_forward(w)
e1.record()
torch.cuda.synchronize()
print(f"{name} plain forward: {e0.elapsed_time(e1) / 3:8.3f} ms", flush=True)
