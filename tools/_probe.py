import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from torch_adjoint import model_torch
from nf_amd.normalizing_flows.layers.coupling_cells import PWQuad
from oracle import flow as oflow
dev = torch.device("cuda:0")
kind, d, P, K, NN, N = "pwquad", 11, 9, 128, [59, 55, 18], 591
torch.manual_seed(1000 * d + K)
cell = PWQuad(d, P, K, list(NN)).double()
with torch.no_grad():
    for p in cell.parameters():
        p.add_(0.25 * torch.randn_like(p))
    for mod in cell.NN:
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.uniform_(0, 0.5)
            mod.running_var.uniform_(0.1, 1.0)
x = torch.rand(N, d + 1, dtype=torch.double)
x[:, -1] += 0.5
x[:, :d] *= (1 - 1e-12)
params = {"c." + k: v.numpy() for k, v in cell.state_dict().items()}
want, upd = {}, {}
y_ref = oflow.cell_forward(x.numpy(), params, "c", kind, P, K, False, upd, want)
cell = cell.to(dev).train(False)
model = torch.nn.Sequential(cell)
model.train(False)
torch.manual_seed(5)
R = torch.randn(N, d + 1, dtype=torch.double, device=dev)
def grads(force_torch):
    for p in cell.parameters():
        p.grad = None
    xx = x.to(dev).clone().requires_grad_(True)
    out = model_torch(model, xx) if force_torch else cell(xx)
    (out * R).sum().backward()
    return xx.grad.cpu().numpy(), out.detach().cpu().numpy()
gh, oh = grads(False)
gt, ot = grads(True)
print("fwd max rel diff hip-torch", (np.abs(oh - ot) / np.maximum(np.abs(ot), 1e-300)).max(), "hip-oracle", (np.abs(oh - y_ref) / np.maximum(np.abs(y_ref), 1e-300)).max(), "max J", np.abs(y_ref[:, -1]).max())
diff = np.abs(gh - gt)
rows = np.where((diff > 1e-6 * np.abs(gt) + 1e-7 * np.maximum(1.0, np.abs(gt).max(axis=1, keepdims=True))).any(axis=1))[0]
print("bad rows", rows[:20], "of", len(rows))
bins = want["bins"]
for r in rows[:6]:
    print("row", r, "bins", bins[r], "x_T", x[r, P:d].numpy(), "J_ref", y_ref[r, -1])
    print("  rowmax gt", np.abs(gt[r]).max(), "max diff", diff[r].max())
    print("  gt", np.array2string(gt[r], precision=3))
    print("  gh", np.array2string(gh[r], precision=3))
print("global max |gt|", np.abs(gt).max(), "argmax row", np.unravel_index(np.abs(gt).argmax(), gt.shape))
