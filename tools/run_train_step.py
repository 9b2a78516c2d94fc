"""Time one training minibatch (forward + backward of var(f*J/maxf)) of the C2 model at mbs = 1e6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from torch_adjoint import model_torch          # test infrastructure: the stock-PyTorch adjoint the kernels are compared with
import torch
from nf_amd import integrands
from nf_amd.normalizing_flows.manager import PWQuadManager, hip_var
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
f = integrands.Camel()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = torch.rand(N, 2, dtype=torch.double, device=dev)
for mode in ("hip", "torch"):
    def step():
        for p in NF._model.parameters():
            p.grad = None
        XJ = model_torch(NF._model, w, has_jac=False) if mode == "torch" else NF._forward(w, dev)   # the latter: what _train_variance_forward_seq calls
        fres = f(XJ[:, :-1].detach()) * XJ[:, -1]
        loss = hip_var(fres)
        loss.backward()
        return loss
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{mode} adjoint: N={N} fwd+bwd {dt*1e3:.3f} ms  {N/dt/1e6:.1f} Msamples/s  loss={float(loss):.6e}", flush=True)
