"""PyTorch-CPU float64 restatement of the PWQuad flow + integrate loop (oracle, test / baseline only).

The same algorithm as oracle/flow.py (NumPy), op for op, on torch CPU tensors, so that it runs on ALL host cores
(torch intra-op threads) -- the "reference-equivalent CPU path" SURVEY.md 8(d) asks bench.py to time beside the GPU number
(the reference itself is PyTorch-CPU code and cannot travel to the GPU box).  Pinned to oracle/flow.py, which is pinned to
the reference's golden vectors: tests/test_oracle_golden.py::test_torch_restatement_matches_numpy.

Reference files restated (paths relative to /root/reference/nisrep/normalizing_flows):
  layers/coupling_cells.py:159-228  PWQuad.forward    -> pwquad_from_Z
  layers/coupling_cells.py:236-254  RectNN            -> conditioner (BatchNorm1d train / eval semantics)
  layers/layers.py:75-77, :90-91    AddJacobian, Roll -> model_forward
  manager.py:393-405                integrate         -> integrate
"""
import math

import torch

from . import flow as _np_flow

BN_EPS = _np_flow.BN_EPS
BN_MOMENTUM = _np_flow.BN_MOMENTUM


def to_torch(params):
    return {k: torch.as_tensor(v, dtype=torch.double) if k.split(".")[-1] != "num_batches_tracked" else torch.as_tensor(v)
            for k, v in params.items()}


def batchnorm(h, params, key, train, update=None):
    """torch.nn.BatchNorm1d forward on (N, F) input (coupling_cells.py:236,238,244)."""
    g, b = params[key + ".weight"], params[key + ".bias"]
    if train:
        n = h.shape[0]
        mean = h.mean(dim=0)
        var = ((h - mean) ** 2).mean(dim=0)
        if update is not None:
            unb = var * (n / (n - 1.0)) if n > 1 else var
            update[key + ".running_mean"] = (1 - BN_MOMENTUM) * params[key + ".running_mean"] + BN_MOMENTUM * mean
            update[key + ".running_var"] = (1 - BN_MOMENTUM) * params[key + ".running_var"] + BN_MOMENTUM * unb
            update[key + ".num_batches_tracked"] = params[key + ".num_batches_tracked"] + 1
    else:
        mean, var = params[key + ".running_mean"], params[key + ".running_var"]
    scale = g / torch.sqrt(var + BN_EPS)
    return h * scale + (b - mean * scale)


def conditioner(xA, params, prefix, train, update=None):
    """BN(P) -> Linear(no bias) -> BN -> ReLU -> ... -> Linear(+bias)   (coupling_cells.py:236-254)."""
    L = _np_flow.n_hidden_from_params(params, prefix)
    bn, lin = _np_flow.bn_index, _np_flow.lin_index
    h = batchnorm(xA, params, f"{prefix}.NN.{bn(0)}", train, update)
    for l in range(L):
        h = h @ params[f"{prefix}.NN.{lin(l)}.weight"].T
        h = batchnorm(h, params, f"{prefix}.NN.{bn(l + 1)}", train, update)
        h = torch.clamp_min(h, 0.0)
    return h @ params[f"{prefix}.NN.{lin(L)}.weight"].T + params[f"{prefix}.NN.{lin(L)}.bias"]


def _lerp(a, b, w):
    """torch.lerp on CPU (used at coupling_cells.py:224)."""
    return torch.where(w < 0.5, a + w * (b - a), b - (b - a) * (1.0 - w))


def pwquad_from_Z(xB, jac, Z, K):
    """Piecewise-quadratic CDF transform given the conditioner output (coupling_cells.py:167-225)."""
    xB = torch.where(xB > 1.0 - 1e-6, torch.full_like(xB, 1.0 - 1e-6), xB)      # :167
    V = torch.exp(Z[:, :, :K + 1])                                              # :173,189
    W = torch.exp(Z[:, :, K + 1:])                                              # :175,178
    Wsum = torch.cumsum(W, dim=-1)                                              # :180
    Wnorm = Wsum[:, :, -1:]
    W = W / Wnorm                                                               # :184
    Wsum = Wsum / Wnorm                                                         # :186
    area = torch.cumsum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, dim=-1)           # :194
    V = V / area[:, :, -1:]                                                     # :196-197
    cand = torch.where(Wsum > xB[:, :, None], torch.zeros_like(Wsum), Wsum)     # :198-202
    aug = torch.cat([torch.full(cand.shape[:2] + (1,), 1e-30, dtype=cand.dtype), cand], dim=-1)
    b = torch.argmax(aug, dim=-1)
    left = torch.cat([torch.zeros_like(Wsum[:, :, :1]), Wsum], dim=-1)          # Wsum2  :198

    def take(A, i):
        return torch.gather(A, -1, i[:, :, None])[:, :, 0]

    Wb = take(W, b)
    alpha = (xB - take(left, b)) / Wb                                           # :206-207
    VW = torch.cat([torch.zeros_like(area[:, :, :1]),
                    torch.cumsum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, dim=-1)], dim=-1)    # :209-210
    Vb, Vb1 = take(V, b), take(V, b + 1)
    yB = (alpha ** 2) / 2 * ((Vb1 - Vb) * Wb) + (alpha * Vb) * Wb + take(VW, b)  # :213-220
    jac = jac * torch.prod(_lerp(Vb, Vb1, alpha), dim=-1)                        # :224-225
    return yB, jac, b


def cell_forward(x, params, prefix, P, K, train, update=None):
    """One PWQuad cell on x = (N, d+1) [d variables | running Jacobian]  (coupling_cells.py:159)."""
    d = x.shape[1] - 1
    T = d - P
    xA, xB, jac = x[:, :P], x[:, P:d], x[:, d]
    Z = conditioner(xA, params, prefix, train, update).reshape(x.shape[0], T, 2 * K + 1)
    yB, jac, _ = pwquad_from_Z(xB, jac, Z, K)
    return torch.cat([xA, yB, jac[:, None]], dim=1)


def model_forward(w, params, steps, train, update=None):
    """AddJacobian (layers.py:75-77) followed by the step list of oracle.flow.plan_pwquad (d <= 7: cells and rolls)."""
    x = torch.cat([w.to(torch.double), torch.ones((w.shape[0], 1), dtype=torch.double)], dim=1)
    d = x.shape[1] - 1
    cur = dict(params)
    for st in steps:
        if st[0] == "cell":
            if st[2] != "pwquad":
                raise NotImplementedError("flow_torch restates the PWQuad model (bench cpu_baseline)")
            upd = {} if train else None
            x = cell_forward(x, cur, st[1], st[3], st[4], train, upd)
            if upd:
                cur.update(upd)
                if update is not None:
                    update.update(upd)
        elif st[0] == "roll":                                                   # layers.py:90-91
            x = torch.cat([torch.roll(x[:, :d], st[1], dims=1), x[:, d:]], dim=1)
        else:
            raise NotImplementedError(st[0])
    return x


def camel(x):
    """README.md:35-36."""
    return torch.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / (0.2 ** 2)) + \
        torch.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / (0.2 ** 2))


def integrate(f, params, steps, w_draws, train=True):
    """BasicManager.integrate (manager.py:393-405) on pre-drawn float32 w batches; returns (sig, sig_err, means, vars, params)."""
    cur = dict(params)
    means, variances = [], []
    for w in w_draws:
        upd = {}
        XJ = model_forward(w, cur, steps, train, upd)
        cur.update(upd)
        fres = f(XJ[:, :-1]) * XJ[:, -1]
        means.append(fres.mean().to(torch.float32))                             # float32 stores (:391-392)
        variances.append(fres.var().to(torch.float32))
    mean, var = torch.stack(means), torch.stack(variances)
    sig = torch.sum(mean / var) / torch.sum(1 / var)                            # :402
    sig_err = torch.sqrt(1 / torch.sum(1 / var)) / math.sqrt(w_draws[0].shape[0] * len(w_draws))   # :403
    return float(sig), float(sig_err), mean, var, cur
