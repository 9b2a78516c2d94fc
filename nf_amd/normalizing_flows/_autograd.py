"""Autograd glue for the HIP flow.

Forward values AND gradients come from the HIP kernels.  Gradients are needed only by `_train_variance_forward_seq`
(/root/reference/nisrep/normalizing_flows/manager.py:275-280): `nis_flow_forward_train` keeps what the adjoint needs,
`nis_flow_backward` runs the hand-written adjoint (nf_amd/csrc/flow_bwd.hip: conditioner widths and pass-through size <= 8 on
the fp64 VALU; flow_bwd_wm.hip: widths 9..64 on the f64 matrix cores; PWQuad n_bins <= 32, PWLin any n_bins whose LDS image
fits; AffineCoupling), with train-mode BatchNorm or -- like the reference's autograd, which differentiates a model in any mode --
eval-mode BatchNorm (running statistics: `nis_flow_forward_hist` / `nis_flow_backward_eval`).  Shapes outside that class RAISE:
there is no stock-PyTorch fallback in the product (the torch re-evaluation the adjoint kernels are tested against lives in
tests/torch_adjoint.py).
"""
import torch

from .. import _lib
from ..flow import _param_tensors, ensure_flat


class _HipFlowFn(torch.autograd.Function):
    """Forward AND backward in libnis_hip.so."""

    @staticmethod
    def forward(ctx, engine, train, has_jac, x, *params):
        xj, saved = engine.forward_train(x.detach(), has_jac, train)
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


class _HipLossFn(torch.autograd.Function):
    """One training minibatch from the raw samples to the scalar loss, forward AND backward in libnis_hip.so
    (/root/reference/nisrep/normalizing_flows/manager.py:220-268): nis_flow_forward_train -> integrand -> nis_loss_terms (one
    reduction pass for the integ / err bookkeeping, the loss and the variance of the mean), and on backward
    nis_flow_backward_loss, whose first launch forms dL/dJ from the records -- no (N, d+1) cotangent, no elementwise torch glue."""

    @staticmethod
    def forward(ctx, engine, w, f, maxf, preburn, est, *params):
        xj, saved = engine.forward_train(w.detach(), False)
        if hasattr(f, "nis_kind"):
            fv, rec, out = engine.loss_terms(xj, maxf, preburn, est, f=f, w=w.detach())
        else:
            with torch.no_grad():
                vals = f(w.detach()) if preburn else f(xj[:, :-1])
            fv, rec, out = engine.loss_terms(xj, maxf, preburn, est, fvals=vals)
        ctx.engine, ctx.saved, ctx.fv, ctx.rec, ctx.maxf, ctx.est = engine, saved, fv, rec, maxf, est
        ctx.sizes = [[p.numel() for p in c.parameters()] for c in engine.cells]
        ctx.shapes = [[p.shape for p in c.parameters()] for c in engine.cells]
        ctx.mark_non_differentiable(out)
        return out[2].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        gflat = ctx.engine.backward_loss(ctx.saved, ctx.fv, ctx.maxf, ctx.rec, gloss, ctx.est)
        grads = []
        for gf, sizes, shapes in zip(gflat, ctx.sizes, ctx.shapes):
            off = 0
            for n, sh in zip(sizes, shapes):
                grads.append(gf[off:off + n].view(sh))
                off += n
        return (None, None, None, None, None, None) + tuple(grads)


class _PlanToken:
    """Lives as long as the autograd node that owns a TrainPlan's current history (flow.TrainPlan.busy)."""
    __slots__ = ("__weakref__",)


class _HipEpochLossFn(torch.autograd.Function):
    """The G minibatches of ONE optimizer step (manager.py:219-270: the epoch's loss is the mean of the minibatch losses and
    backward runs once) as three library calls: nis_flow_forward_hist_batched -> nis_loss_terms_batched -> (backward)
    nis_flow_backward_loss_batched.  The minibatches are the y dimension of every launch.  Returns (sum of the G minibatch
    losses, out (G, 8) bookkeeping rows)."""

    @staticmethod
    def forward(ctx, engine, w, f, maxf, preburn, est, *params):
        w = w.detach()
        builtin = hasattr(f, "nis_kind")
        ctx.plan = None
        plan = engine.train_plan(w.shape[0], w.shape[1], w.device, f, maxf, preburn, est) if builtin and _train_graph_ok(engine, w) else None
        # the plan has ONE history: while an earlier forward of the same shape still awaits its backward (an epoch of more minibatches
        # than one call carries: manager.py:212-280 runs every chunk's forward before the single loss.backward()) this call runs eagerly
        if plan is not None and not plan.busy():
            # launch-bound regime: both halves of the step are captured graphs (flow.TrainPlan); same calls, same results
            ctx.token = _PlanToken()
            out = plan.forward(w, holder=ctx.token)
            ctx.engine, ctx.plan, ctx.generation = engine, plan, plan.generation
            ctx.mark_non_differentiable(out)
            return out[:, 2].sum(), out
        xj, saved = engine.forward_train_batched(w, True, want_xj=not builtin and not preburn)
        if builtin:
            fv, rec, out = engine.loss_terms_batched(saved, maxf, preburn, est, f=f, w=w)
        else:
            with torch.no_grad():               # the callable sees one minibatch at a time, as in the reference's loop
                vals = torch.stack([f(w[g]) if preburn else f(xj[g, :, :-1]) for g in range(w.shape[0])])
            fv, rec, out = engine.loss_terms_batched(saved, maxf, preburn, est, fvals=vals)
        ctx.engine, ctx.saved, ctx.fv, ctx.rec, ctx.maxf, ctx.est = engine, saved, fv, rec, maxf, est
        ctx.mark_non_differentiable(out)
        return out[:, 2].sum(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        if ctx.plan is not None:
            gflat = ctx.plan.backward(gloss, ctx.generation)
            ctx.token = None                    # the history is free again
        else:
            gflat = ctx.engine.backward_loss_batched(ctx.saved, ctx.fv, ctx.maxf, ctx.rec, gloss, ctx.est)
            ctx.saved = ctx.fv = None
        grads = []
        for gf, c in zip(gflat, ctx.engine.cells):
            off = 0
            for p in _param_tensors(c)[0]:
                n = p.numel()
                grads.append(gf[off:off + n].view(p.shape))
                off += n
        return (None, None, None, None, None, None) + tuple(grads)


TRAIN_GRAPH_MAX_SAMPLES = 1 << 21      # per optimizer step: above this the launches are long and eager submission keeps up
TRAIN_GRAPH = False                    # opt-in (or NIS_TRAIN_GRAPH=1): see _train_graph_ok


def _train_graph_ok(engine, w):
    """Captured training step (flow.TrainPlan): OPT-IN.  Measured on one MI355X: 0.88 -> 0.74 ms per step of 5 x 2 000, 0.93 -> 0.62 ms
    for 1 x 10 000, 0.90 -> 0.73 ms for 1 x 1e6 (C2 model), nothing for matrix-core models (their chain is device-latency bound) --
    but capturing the two graphs costs ~0.2 s per (set size, preburn, loss form): the reference's README run (300 epochs) goes from
    0.61 to 0.99 s.  It pays for trainings of thousands of epochs, hence the switch."""
    import os
    if not (TRAIN_GRAPH or os.environ.get("NIS_TRAIN_GRAPH", "0") not in ("", "0")):
        return False
    if w.dtype != torch.double or w.shape[0] * w.shape[1] > TRAIN_GRAPH_MAX_SAMPLES:
        return False
    return not torch.cuda.is_current_stream_capturing()


def flow_loss_batched(model, w, f, maxf, preburn, est):
    """(sum of the minibatch losses: 0-d tensor with grad, out (G, 8) rows {mean(a), var(a), loss, var(u^2), n, mean(u), -, -}) for
    the minibatches w (G, N, d), or None when the model is outside the class the fused path covers."""
    engine = model.engine()
    if not (model.training and engine.backward_supported() and w.is_cuda and w.dtype == torch.double and torch.is_grad_enabled()):
        return None
    for c in engine.cells:
        ensure_flat(c, w.device)
    params = [p for c in engine.cells for p in _param_tensors(c)[0]]        # = c.parameters() order (the blob order)
    return _HipEpochLossFn.apply(engine, w, f, maxf, preburn, est, *params)


def flow_loss(model, w, f, maxf, preburn, est):
    """(loss 0-d tensor with grad, out8 device tensor {mean(a), var(a), loss, var(u^2), n, mean(u)}) of one minibatch, or None
    when the model is outside the class the fused path covers (then the caller runs the piecewise path)."""
    engine = model.engine()
    if not (model.training and engine.backward_supported() and w.is_cuda and w.dtype == torch.double and torch.is_grad_enabled()):
        return None
    from ..flow import ensure_flat
    for c in engine.cells:
        ensure_flat(c, w.device)
    params = [p for c in engine.cells for p in c.parameters()]
    return _HipLossFn.apply(engine, w, f, maxf, preburn, est, *params)


class _ForwardOnlyFn(torch.autograd.Function):
    """HIP forward whose backward refuses: the reference lets a model be CALLED in eval mode (or with any cell shape) while
    autograd records; only asking for the gradient of a case the adjoint kernels do not cover is an error here."""

    @staticmethod
    def forward(ctx, engine, train, has_jac, why, x, *params):
        ctx.why = why
        return engine.forward(x.detach(), has_jac, train, want_xj=True)["xj"]

    @staticmethod
    def backward(ctx, gout):
        raise _lib.NisError(ctx.why)


def run_flow(engine, modules, x, has_jac, train):
    """Model / cell forward through the HIP engine; differentiable when autograd is recording."""
    cells = [m for m in modules if hasattr(m, "KIND")]
    params = [p for c in cells for p in c.parameters()]
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need:
        return engine.forward(x, has_jac, train, want_xj=True)["xj"]
    why = None
    if not engine.backward_supported():
        why = ("the adjoint kernels do not cover this model (a matrix-core head whose one-column LDS ring does not fit): forward-only "
               "on this build")
    if why is not None and train:
        # a TRAINING forward outside the adjoint class: refuse before anything runs -- the forward would advance the BatchNorm
        # running statistics and the failure would surface mid-step, inside loss.backward()
        raise _lib.NisError(why)
    if why is not None:
        return _ForwardOnlyFn.apply(engine, train, has_jac, why, x, *params)
    from ..flow import ensure_flat
    for c in cells:
        ensure_flat(c, x.device)        # before autograd captures the parameter tensors
    params = [p for c in cells for p in c.parameters()]
    return _HipFlowFn.apply(engine, train, has_jac, x, *params)      # frozen parameters: their gradients are computed and dropped
