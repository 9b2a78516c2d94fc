"""Stock-PyTorch re-evaluation of the coupling cells (TEST INFRASTRUCTURE, like oracle/): the same maths as the HIP forward
written with differentiable torch ops, so that autograd through it gives an independent adjoint to compare the hand-written
adjoint kernels (nf_amd/csrc/flow_bwd*.hip) with.  Itself checked against the HIP forward (output values) in every test that
uses it, and the HIP adjoint is additionally pinned to the REFERENCE's autograd (tests/golden/training.npz, wide.npz).

Restates /root/reference/nisrep/normalizing_flows/layers/coupling_cells.py:56-68 (affine), :107-142 (PWLin), :159-228
(PWQuad), :236-254 (conditioner).  The product package does not contain, import or call this file.
"""
import math

import torch
import torch.nn.functional as F

from nf_amd import _lib


def _bn_train(h, bn):
    # batch statistics, no running-stat side effect (the HIP forward already applied it)
    return F.batch_norm(h, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)


def _bn_eval(h, bn):
    return F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


def _conditioner_torch(cell, xA, train):
    h = xA
    for m in cell.NN:
        if isinstance(m, torch.nn.BatchNorm1d):
            h = _bn_train(h, m) if train else _bn_eval(h, m)
        elif isinstance(m, torch.nn.Linear):
            h = F.linear(h, m.weight, m.bias)
        elif isinstance(m, torch.nn.ReLU):
            h = torch.relu(h)
    if cell.KIND == _lib.NIS_AFFINE:
        return h.reshape(h.shape[0], 2, cell.transform_size)
    return h.reshape(h.shape[0], cell.transform_size, -1)


def _gather(A, idx):
    return torch.gather(A, -1, idx.unsqueeze(-1)).squeeze(-1)


def cell_torch(cell, x, train):
    """Differentiable re-evaluation of one cell (same maths as k_cell_apply) for the backward pass."""
    P, d, K = cell.pass_through_size, cell.flow_size, getattr(cell, "n_bins", 0)
    xA, xB, jac = x[:, :P], x[:, P:d], x[:, d]
    Z = _conditioner_torch(cell, xA, train)
    if cell.KIND == _lib.NIS_AFFINE:
        s0 = 20 * torch.exp(Z[:, 0])
        u = xB * s0 + torch.relu(Z[:, 1])
        yB = torch.atan(u) / (math.pi / 2)
        jac = jac * torch.prod(s0, 1) * (1 / (math.pi / 2)) * torch.prod(1 / (u ** 2 + 1), 1)
    elif cell.KIND == _lib.NIS_PWQUAD:
        xB = torch.clamp(xB, max=1.0 - 1e-6)
        V = torch.exp(Z[:, :, :K + 1])
        W = torch.exp(Z[:, :, K + 1:])
        Wc = torch.cumsum(W, -1)
        Wn = Wc[:, :, -1:]
        W = W / Wn
        Wc = Wc / Wn
        area = torch.sum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, -1, keepdim=True)
        V = V / area
        with torch.no_grad():
            cand = torch.where(Wc > xB.unsqueeze(-1), torch.zeros_like(Wc), Wc)
            aug = torch.cat((torch.full_like(cand[:, :, :1], 1e-30), cand), -1)
            b = torch.argmax(aug, -1)
        left = _gather(torch.cat((torch.zeros_like(Wc[:, :, :1]), Wc), -1), b)
        Wb = _gather(W, b)
        alpha = (xB - left) / Wb
        VW = torch.cat((torch.zeros_like(area), torch.cumsum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, -1)), -1)
        Vb, Vb1 = _gather(V, b), _gather(V, b + 1)
        yB = (alpha ** 2) / 2 * ((Vb1 - Vb) * Wb) + (alpha * Vb) * Wb + _gather(VW, b)
        jac = jac * torch.prod(torch.lerp(Vb, Vb1, alpha), -1)
    else:
        Q = torch.exp(Z)
        Qc = torch.cumsum(Q, -1)
        Qn = Qc[:, :, -1:]
        Q = Q / (Qn / K)
        Qc = torch.cat((torch.zeros_like(Qn), Qc / Qn), -1)
        a = xB * K
        bf = torch.floor(a)
        alpha = (a - bf) / K
        b = bf.long()
        qb = _gather(Q, b)
        yB = qb * alpha + _gather(Qc, b)
        jac = jac * torch.prod(qb, -1)
    return torch.cat((xA, yB, jac.unsqueeze(-1)), -1)


def modules_torch(modules, x, train):
    for m in modules:
        x = cell_torch(m, x, train) if hasattr(m, "KIND") else m(x)
    return x




def model_torch(model, x, has_jac=True):
    """Differentiable XJ of a FlowSequential (or any iterable of cells / permutation layers) on the model's device."""
    train = model.training
    xj = x if has_jac else torch.cat((x, torch.ones_like(x[:, :1])), -1)
    return modules_torch(list(model.children()), xj, train)
