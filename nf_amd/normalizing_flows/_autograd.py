"""Autograd glue for the HIP flow.

Forward values ALWAYS come from the HIP kernels.  Gradients (needed only by
`_train_variance_forward_seq`, /root/reference/nisrep/normalizing_flows/manager.py:275-280):
  * models whose cells are all in the class the adjoint kernels cover (nf_amd/csrc/flow_bwd.hip: conditioner widths and
    pass-through size <= 8 on the fp64 VALU; flow_bwd_wm.hip: widths 9..64 on the f64 matrix cores; PWQuad n_bins <= 32,
    PWLin any n_bins whose LDS image fits) use the hand-written HIP backward (`nis_flow_forward_train` +
    `nis_flow_backward`);
  * every other model (a PWLin head whose LDS image does not fit) re-evaluates the same cells with stock PyTorch-ROCm ops on the GPU inside
    `backward` and differentiates that graph.  That
    recomputation is a device-side adjoint helper, not a fallback: it never produces a forward result and refuses to run
    without a GPU.  NF_AMD_TORCH_ADJOINT=1 forces it everywhere (A/B checks).
"""
import math
import os

import torch
import torch.nn.functional as F

from .. import _lib


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


class _HipFlowFn(torch.autograd.Function):
    """Forward AND backward in libnis_hip.so."""

    @staticmethod
    def forward(ctx, engine, train, has_jac, x, *params):
        xj, saved = engine.forward_train(x.detach(), has_jac)
        ctx.engine, ctx.saved = engine, saved
        ctx.sizes = [[p.numel() for p in c.parameters()] for c in engine.cells]
        ctx.shapes = [[p.shape for p in c.parameters()] for c in engine.cells]
        ctx.x_dtype = x.dtype
        return xj

    @staticmethod
    def backward(ctx, gout):
        gflat, gx = ctx.engine.backward(ctx.saved, gout, ctx.needs_input_grad[3])
        grads = []
        for gf, sizes, shapes in zip(gflat, ctx.sizes, ctx.shapes):
            off = 0
            for n, sh in zip(sizes, shapes):
                grads.append(gf[off:off + n].view(sh))
                off += n
        if gx is not None:
            if not ctx.saved[2]:
                gx = gx[:, :-1]
            gx = gx.to(ctx.x_dtype)
        return (None, None, None, gx) + tuple(grads)


class _FlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, modules, has_jac, train, x, *params):
        out = engine.forward(x.detach(), has_jac, train, want_xj=True)
        ctx.modules, ctx.has_jac, ctx.train = modules, has_jac, train
        ctx.save_for_backward(x)
        ctx.n_params = len(params)
        return out["xj"]

    @staticmethod
    def backward(ctx, gout):
        _lib.require_gpu()
        (x,) = ctx.saved_tensors
        params = [p for m in ctx.modules if hasattr(m, "KIND") for p in m.parameters()]
        with torch.enable_grad():
            xin = x.detach().to(torch.double).requires_grad_(ctx.needs_input_grad[4])
            xj_in = xin if ctx.has_jac else torch.cat((xin, torch.ones_like(xin[:, :1])), -1)
            y = modules_torch(ctx.modules, xj_in, ctx.train)
            wanted = ([xin] if ctx.needs_input_grad[4] else []) + [p for p in params if p.requires_grad]
            grads = torch.autograd.grad(y, wanted, gout, allow_unused=True)
        grads = list(grads)
        gx = grads.pop(0) if ctx.needs_input_grad[4] else None
        if gx is not None and gx.dtype != x.dtype:
            gx = gx.to(x.dtype)
        gp = []
        for p in params:
            gp.append(grads.pop(0) if p.requires_grad else None)
        return (None, None, None, None, gx) + tuple(gp)


def run_flow(engine, modules, x, has_jac, train):
    """Model / cell forward through the HIP engine; differentiable when autograd is recording."""
    cells = [m for m in modules if hasattr(m, "KIND")]
    params = [p for c in cells for p in c.parameters()]
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need:
        return engine.forward(x, has_jac, train, want_xj=True)["xj"]
    from ..flow import ensure_flat
    for c in cells:
        ensure_flat(c, x.device)        # before autograd captures the parameter tensors
    params = [p for c in cells for p in c.parameters()]
    if train and engine.backward_supported() and not os.environ.get("NF_AMD_TORCH_ADJOINT") \
            and all(p.requires_grad for p in params):
        return _HipFlowFn.apply(engine, train, has_jac, x, *params)
    return _FlowFn.apply(engine, modules, has_jac, train, x, *params)
