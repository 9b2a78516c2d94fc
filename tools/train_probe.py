"""Time the fused training step: G minibatches of mbs samples per optimizer step (forward with history, integrand, loss
bookkeeping, adjoint, Adamax), per-minibatch entry points against the batched ones (grid.y).

  python tools/train_probe.py [config] [mbs] [G] [steps] [mode]     config: C2 | C3 | C4 | C5; mode: both | batched | single
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
from nf_amd.normalizing_flows._autograd import flow_loss, flow_loss_batched

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
mbs = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
mode = sys.argv[5] if len(sys.argv) > 5 else "both"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if cfg == "C2":
    NF = PWQuadManager(n_flow=2); NF.create_model(2, 4, [3] * 3); d = 2; f = integrands.Camel()
elif cfg == "C3":
    NF = PWLinManager(n_flow=8); NF.create_model(4, 4, 64, [32] * 3, 4); d = 8; f = integrands.Normal(0.5, 0.25, 8)
elif cfg == "C4":
    NF = PWQuadManager(n_flow=10); NF.create_model(5, 16, [32] * 3); d = 10; f = integrands.Normal(0.5, 0.25, 10)
else:
    NF = PWQuadManager(n_flow=16); NF.create_model(8, 16, [32] * 3); d = 16; f = integrands.Normal(0.5, 0.25, 16)
model = NF._model.to(dev).train()
optim = torch.optim.Adamax(model.parameters(), lr=2e-3, weight_decay=1e-4)
maxf = torch.ones((), dtype=torch.double, device=dev)


def step_single():
    optim.zero_grad()
    loss = 0
    for _ in range(G):
        w = torch.rand(mbs, d, dtype=torch.double, device=dev)
        l, _t = flow_loss(model, w, f, maxf, False, False)
        loss = loss + l
    (loss / G).backward()
    optim.step()
    return loss / G


def step_batched():
    optim.zero_grad()
    w = torch.rand(G, mbs, d, dtype=torch.double, device=dev)
    ls, _t = flow_loss_batched(model, w, f, maxf, False, False)
    loss = ls / G
    loss.backward()
    optim.step()
    return loss


for name, fn in (("single", step_single), ("batched", step_batched)):
    if mode not in ("both", name):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        loss = fn()
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{cfg} {name:8s} G={G} mbs={mbs}: {dt * 1e3:8.3f} ms/step wall, {e0.elapsed_time(e1) / steps:8.3f} ms/step stream, "
          f"{G * mbs / dt / 1e6:9.1f} Msamples/s  loss={float(loss.detach()):.6e}", flush=True)
