"""Host time of the sections of one fused training step (no GPU sync inside: pure launch-side cost).  python tools/host_sections.py [mbs] [G] [C5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWQuadManager
from nf_amd.normalizing_flows._autograd import flow_loss_batched
mbs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
if len(sys.argv) > 3 and sys.argv[3] == "C5":
    NF = PWQuadManager(n_flow=16); NF.create_model(8, 16, [32] * 3); f = integrands.Normal(0.5, 0.25, 16)
else:
    NF = PWQuadManager(n_flow=2); NF.create_model(2, 4, [3] * 3); f = integrands.Camel()
dflow = NF.n_flow
model = NF._model.to(dev).train()
optim = torch.optim.Adamax(model.parameters(), lr=2e-3, weight_decay=1e-4)
maxf = torch.ones((), dtype=torch.double, device=dev)
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
for it in range(220):
    if it == 20:
        torch.cuda.synchronize(); acc.clear(); T0 = time.perf_counter()
    t = time.perf_counter()
    optim.zero_grad(); t = tick("zero_grad", t)
    w = torch.rand(G, mbs, dflow, dtype=torch.double, device=dev); t = tick("rand", t)
    ls, _t = flow_loss_batched(model, w, f, maxf, False, False); t = tick("forward+loss", t)
    loss = ls / G; t = tick("div", t)
    loss.backward(); t = tick("backward", t)
    optim.step(); t = tick("optim.step", t)
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(f"mbs={mbs} G={G}: {tot / 200 * 1e3:.3f} ms/step wall")
for k, v in acc.items():
    print(f"  {k:14s} {v / 200 * 1e6:8.1f} us")
