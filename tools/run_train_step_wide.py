"""Time one training minibatch (forward + backward) of the wide configurations C3 / C4 / C5:
hand-written matrix-core adjoint against the stock-PyTorch adjoint.  usage: run_train_step_wide.py [N] [configs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from torch_adjoint import model_torch          # test infrastructure: the stock-PyTorch adjoint the kernels are compared with
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager, hip_var
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3", "C4", "C5"]
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["hip", "torch"]


def build(name):
    torch.manual_seed(0)
    if name == "C3":
        NF = PWLinManager(n_flow=8)
        NF.create_model(4, 4, 64, [32] * 3, 4)
        return NF, 8
    if name == "C4":
        NF = PWQuadManager(n_flow=10)
        NF.create_model(5, 16, [32] * 3)
        return NF, 10
    NF = PWQuadManager(n_flow=16)
    NF.create_model(8, 16, [32] * 3)
    return NF, 16


for name in which:
    NF, d = build(name)
    f = integrands.Normal(0.5, 0.1, d) if hasattr(integrands, "Normal") else None
    w = torch.rand(N, d, dtype=torch.double, device=dev) * (1 - 1e-9)
    for mode in modes:
        def step():
            for p in NF._model.parameters():
                p.grad = None
            XJ = model_torch(NF._model, w, has_jac=False) if mode == "torch" else NF._forward(w, dev)   # the latter: what _train_variance_forward_seq calls
            fres = torch.exp(-((XJ[:, :-1].detach() - 0.5) ** 2).sum(-1) / 0.5) * XJ[:, -1]
            loss = hip_var(fres)
            loss.backward()
            return loss

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        gn = sum(float(p.grad.abs().sum()) for p in NF._model.parameters())
        print(f"{name} {mode} adjoint: N={N} fwd+bwd {dt*1e3:.2f} ms  {N/dt/1e6:.2f} Msamples/s  loss={float(loss):.6e} |grad|_1={gn:.6e}",
              flush=True)
