"""NumPy float64 restatement of the coupling-cell flow of NGoetz/NF (oracle, test-only).

Reference files restated (paths relative to /root/reference/nisrep/normalizing_flows):
  layers/coupling_cells.py:73-142   PWLin      -> pwlin_cell
  layers/coupling_cells.py:144-228  PWQuad     -> pwquad_cell
  layers/coupling_cells.py:6-70     AffineCoupling -> affine_from_Z (conditioner with hidden-layer biases :26-40)
  layers/coupling_cells.py:230-254  RectNN     -> conditioner  (same stack inline in PWLin :84-104)
  layers/layers.py:6-51,66-91       Mask/DeMask/AddJacobian/Roll -> plan steps in model_forward
  manager.py:474-499, 518-600       create_model topology -> plan_pwlin / plan_pwquad
  manager.py:428-453                AffineManager.create_model -> plan_affine
  manager.py:380-405                integrate  -> integrate_combine (+ moments)

Parameters are passed as a dict of NumPy arrays keyed like the reference's
`state_dict()` ("<cell>.NN.<idx>.<weight|bias|running_mean|running_var|num_batches_tracked>").
Module indices inside `NN`: BN_0 at 0, Linear_l at 1+3l (l = 0..L), BN_l at 3l-1 (l >= 1).
"""
import math
import numpy as np

BN_EPS = 1e-5        # torch.nn.BatchNorm1d default eps
BN_MOMENTUM = 0.1    # torch.nn.BatchNorm1d default momentum


# ----------------------------------------------------------------------------- conditioner
def bn_index(l):
    return 0 if l == 0 else 3 * l - 1


def lin_index(l):
    return 1 + 3 * l


def n_hidden_from_params(params, prefix):
    L = 0
    while f"{prefix}.NN.{lin_index(L + 1)}.weight" in params:
        L += 1
    return L


def batchnorm(h, params, key, train, update=None):
    """torch.nn.BatchNorm1d forward on (N, F) input (coupling_cells.py:236,238,244).

    train: batch mean / biased variance; running stats get momentum-0.1 update with the
    unbiased variance.  eval: running stats.  `update` (dict) receives the new buffers.
    """
    g = params[key + ".weight"]
    b = params[key + ".bias"]
    if train:
        n = h.shape[0]
        mean = h.mean(axis=0)
        var = ((h - mean) ** 2).mean(axis=0)
        if update is not None:
            unb = var * (n / (n - 1.0)) if n > 1 else var
            update[key + ".running_mean"] = (1 - BN_MOMENTUM) * params[key + ".running_mean"] + BN_MOMENTUM * mean
            update[key + ".running_var"] = (1 - BN_MOMENTUM) * params[key + ".running_var"] + BN_MOMENTUM * unb
            update[key + ".num_batches_tracked"] = params[key + ".num_batches_tracked"] + 1
    else:
        mean = params[key + ".running_mean"]
        var = params[key + ".running_var"]
    scale = g / np.sqrt(var + BN_EPS)
    return h * scale + (b - mean * scale)


def conditioner(xA, params, prefix, train, update=None):
    """BN(P) -> Linear(no bias) -> BN -> ReLU -> ... -> Linear(+bias)   (coupling_cells.py:236-254).
    AffineCoupling builds the same stack with biased hidden Linears (coupling_cells.py:26-40)."""
    L = n_hidden_from_params(params, prefix)
    h = batchnorm(xA, params, f"{prefix}.NN.{bn_index(0)}", train, update)
    for l in range(L):
        h = h @ params[f"{prefix}.NN.{lin_index(l)}.weight"].T
        if f"{prefix}.NN.{lin_index(l)}.bias" in params:
            h = h + params[f"{prefix}.NN.{lin_index(l)}.bias"]
        h = batchnorm(h, params, f"{prefix}.NN.{bn_index(l + 1)}", train, update)
        h = np.maximum(h, 0.0)
    return h @ params[f"{prefix}.NN.{lin_index(L)}.weight"].T + params[f"{prefix}.NN.{lin_index(L)}.bias"]


# ----------------------------------------------------------------------------- bijectors
def _lerp(a, b, w):
    """torch.lerp on CPU: a + w(b-a) for w < 0.5, else b - (b-a)(1-w)  (used at coupling_cells.py:224)."""
    return np.where(w < 0.5, a + w * (b - a), b - (b - a) * (1.0 - w))


def pwquad_from_Z(xB, jac, Z, K):
    """Piecewise-quadratic CDF transform given the conditioner output (coupling_cells.py:167-225).

    xB (N,T), jac (N,), Z (N,T,2K+1).  Returns yB (N,T), new jac (N,), bin index (N,T).
    """
    xB = np.where(xB > 1.0 - 1e-6, 1.0 - 1e-6, xB)                       # :167
    V = np.exp(Z[:, :, :K + 1])                                            # :173,189
    W = np.exp(Z[:, :, K + 1:])                                            # :175,178
    Wsum = np.cumsum(W, axis=-1)                                           # :180
    Wnorm = Wsum[:, :, -1:]
    W = W / Wnorm                                                          # :184
    Wsum = Wsum / Wnorm                                                    # :186
    area = np.cumsum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, axis=-1)        # :194
    V = V / area[:, :, -1:]                                                # :196-197
    # bin = argmax over [1e-30, Wsum_k if Wsum_k <= xB else 0]  (first maximum)   :198-202
    cand = np.where(Wsum > xB[:, :, None], 0.0, Wsum)
    aug = np.concatenate([np.full(cand.shape[:2] + (1,), 1e-30), cand], axis=-1)
    b = np.argmax(aug, axis=-1)
    left = np.concatenate([np.zeros_like(Wsum[:, :, :1]), Wsum], axis=-1)   # Wsum2  :198
    take = lambda A, i: np.take_along_axis(A, i[:, :, None], axis=-1)[:, :, 0]
    Wb = take(W, b)
    alpha = (xB - take(left, b)) / Wb                                      # :206-207
    VW = np.concatenate([np.zeros_like(area[:, :, :1]),
                         np.cumsum((V[:, :, :-1] + V[:, :, 1:]) / 2 * W, axis=-1)], axis=-1)   # :209-210
    Vb, Vb1 = take(V, b), take(V, b + 1)
    yB = (alpha ** 2) / 2 * ((Vb1 - Vb) * Wb) + (alpha * Vb) * Wb + take(VW, b)    # :213-220
    jac = jac * np.prod(_lerp(Vb, Vb1, alpha), axis=-1)                    # :224-225
    return yB, jac, b


def pwlin_from_Z(xB, jac, Z, K):
    """Piecewise-linear CDF transform with equal-width bins (coupling_cells.py:115-141)."""
    Q = np.exp(Z)                                                          # :115
    Qsum = np.cumsum(Q, axis=-1)                                           # :117
    Qnorm = Qsum[:, :, -1:]
    Q = Q / (Qnorm / K)                                                    # :121
    Qsum = Qsum / Qnorm                                                    # :123
    Qsum = np.concatenate([np.zeros_like(Qsum[:, :, :1]), Qsum], axis=-1)  # :124
    a = xB * K                                                             # :126
    binf = np.floor(a)                                                     # :128
    alpha = (a - binf) / K                                                 # :130-131
    b = binf.astype(np.int64)                                              # :133
    if (b < 0).any() or (b >= K).any():
        raise IndexError("PWLin bin index out of range (reference: torch.gather raises, coupling_cells.py:136)")
    take = lambda A, i: np.take_along_axis(A, i[:, :, None], axis=-1)[:, :, 0]
    qb = take(Q, b)
    yB = qb * alpha + take(Qsum, b)                                        # :136-139
    jac = jac * np.prod(qb, axis=-1)                                       # :141
    return yB, jac, b


def affine_from_Z(xB, jac, Z):
    """Affine coupling transform (coupling_cells.py:56-68).  Z (N,2,T): row 0 -> scale, row 1 -> shift."""
    s0 = np.exp(Z[:, 0])                                                   # :56
    s1 = np.where(Z[:, 1] > 0, Z[:, 1], 0.0)                               # :57
    u = xB * (20 * s0) + s1                                                # :60
    diff = 1 / (u ** 2 + 1)                                                # :62
    yB = np.arctan(u) / (np.pi / 2)                                        # :64
    jac = jac * np.prod(20 * s0, axis=1) * (1 / (np.pi / 2)) * np.prod(diff, axis=1)   # :68 (2/pi enters ONCE)
    return yB, jac


def cell_forward(x, params, prefix, kind, P, K, train, update=None, want=None):
    """One coupling cell on x = (N, d+1) [d variables | running Jacobian]  (coupling_cells.py:107,159)."""
    d = x.shape[1] - 1
    T = d - P
    xA, xB, jac = x[:, :P], x[:, P:d], x[:, d]
    Z = conditioner(xA, params, prefix, train, update)
    if kind == "affine":
        Z = Z.reshape(x.shape[0], 2, T)                                    # Reshape(2, T), coupling_cells.py:44
        yB, jac = affine_from_Z(xB, jac, Z)
        if want is not None:
            want["Z"] = Z
        return np.concatenate([xA, yB, jac[:, None]], axis=1)
    out = (2 * K + 1) if kind == "pwquad" else K
    Z = Z.reshape(x.shape[0], T, out)                                      # Reshape layer, layers.py:63-64
    if kind == "pwquad":
        yB, jac, b = pwquad_from_Z(xB, jac, Z, K)
    else:
        yB, jac, b = pwlin_from_Z(xB, jac, Z, K)
    if want is not None:
        want["Z"], want["bins"] = Z, b
    return np.concatenate([xA, yB, jac[:, None]], axis=1)


# ----------------------------------------------------------------------------- topology
def _bits(x, n):
    """manager.py:20-36 get_bin: binary digits of x, zero-filled to n, most significant first."""
    return [int(c) for c in format(x, "b").zfill(n)]


def pwquad_n_cells(d, n_cells):
    """Automatic adjustment of the number of cells (manager.py:526-534)."""
    if n_cells < 2 * math.ceil(math.log2(d)) and n_cells < d:
        if d <= 6:
            return d
        if d == 7:
            return 6
        return int(2 * math.ceil(math.log2(d)))
    return n_cells


def mask_sets(d, pos):
    """Index sets of MaskLayer(dims_bin, pos) (layers.py:15-24): bit floor(pos/2) of the dimension
    index equals pos%2 -> pass-through ("feeder"), else transformed ("trafoer")."""
    n = len(_bits(d - 1, 0))
    feed, bit = pos % 2, pos // 2
    table = [_bits(i, n) for i in range(d)]
    feeder = [i for i in range(d) if table[i][bit] == feed]
    trafoer = [i for i in range(d) if table[i][bit] != feed]
    return feeder, trafoer


def plan_pwquad(d, n_cells, K):
    """Layer sequence PWQuadManager.create_model builds (manager.py:518-586) as a list of steps:
    ("mask", feeder, trafoer) | ("cell", name, "pwquad", P, K) | ("demask", order) | ("roll", shift)."""
    n_cells = pwquad_n_cells(d, n_cells)
    steps = []
    if d <= 7:
        P = 1 if d <= 6 else 2
        for i in range(n_cells):
            steps.append(("cell", str(i), "pwquad", P, K))
            steps.append(("roll", 1 if i < n_cells - 1 else d - ((n_cells - 1) % d)))
    else:
        n = len(_bits(d - 1, 0))
        for i in range(2 * n):
            feeder, trafoer = mask_sets(d, i)
            steps.append(("mask", feeder, trafoer))
            steps.append(("cell", str(i), "pwquad", len(feeder), K))
            steps.append(("demask", feeder + trafoer))
        rest = n_cells - 2 * n
        for i in range(rest):
            steps.append(("cell", str(i + 2 * n), "pwquad", d // 2, K))
            steps.append(("roll", 1 if i < rest - 1 else d - ((rest - 1) % d)))
    return steps


def plan_pwlin(d, P, n_cells, K, roll_step):
    """PWLinManager.create_model (manager.py:487-492): every roll is registered under the same
    name "roll", so only ONE RollLayer survives, right after cell 0."""
    steps = []
    for i in range(n_cells):
        steps.append(("cell", str(i), "pwlin", P, K))
        if i == 0:
            steps.append(("roll", roll_step))
    return steps


def plan_affine(d, P, n_cells, roll_step):
    """AffineManager.create_model (manager.py:440-446): like PWLinManager, every roll is registered as "roll"."""
    steps = []
    for i in range(n_cells):
        steps.append(("cell", str(i), "affine", P, 0))
        if i == 0:
            steps.append(("roll", roll_step))
    return steps


def model_forward(w, params, steps, train, update=None):
    """AddJacobian (layers.py:75-77) followed by the step list.  w (N,d) -> XJ (N,d+1)."""
    x = np.concatenate([np.asarray(w, dtype=np.float64), np.ones((w.shape[0], 1))], axis=1)
    d = x.shape[1] - 1
    cur = dict(params)
    for st in steps:
        if st[0] == "cell":
            upd = {} if train else None
            x = cell_forward(x, cur, st[1], st[2], st[3], st[4], train, upd)
            if upd:
                cur.update(upd)
                if update is not None:
                    update.update(upd)
        elif st[0] == "roll":                                            # layers.py:90-91
            x = np.concatenate([np.roll(x[:, :d], st[1], axis=1), x[:, d:]], axis=1)
        elif st[0] == "mask":                                            # layers.py:27-32
            x = np.concatenate([x[:, st[1]], x[:, st[2]], x[:, d:]], axis=1)
        elif st[0] == "demask":                                          # layers.py:43-51
            y = np.empty_like(x)
            y[:, st[1]] = x[:, :d]
            y[:, d] = x[:, d]
            x = y
    return x


# ----------------------------------------------------------------------------- integrands / estimators
def camel(x):
    """README.md:35-36."""
    return np.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / (0.2 ** 2)) + \
        np.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / (0.2 ** 2))


def normal(x, mu, sigma, n_flow):
    """manager.py:39-40 (sigma is NOT raised to the power n_flow in the reference; kept)."""
    return np.exp(-np.sum((x - mu) ** 2 / (2 * sigma ** 2), -1)) / (sigma * np.sqrt((2 * np.pi) ** n_flow))


def moments(v):
    """(count, mean, unbiased variance, max) as torch.mean / torch.var / torch.max give (manager.py:399-400)."""
    v = np.asarray(v, dtype=np.float64)
    n = v.shape[0]
    m = v.mean()
    var = ((v - m) ** 2).sum() / (n - 1) if n > 1 else float("nan")
    return n, m, var, v.max()


def integrate_combine(mean, var, neval):
    """manager.py:391-392,402-403: per-iteration mean/var are stored in float32 tensors, combined as
    sig = sum(mean/var)/sum(1/var), sig_err = sqrt(1/sum(1/var))/sqrt(neval*nitn)."""
    mean = np.asarray(mean, dtype=np.float32)
    var = np.asarray(var, dtype=np.float32)
    nitn = mean.shape[0]
    sig = np.float32((mean / var).sum(dtype=np.float32) / (np.float32(1) / var).sum(dtype=np.float32))
    sig_err = np.float32(np.sqrt(np.float32(1) / (np.float32(1) / var).sum(dtype=np.float32))) / np.sqrt(neval * nitn)
    return sig, np.float32(sig_err)


def integrate(f, params, steps, w_draws, train=True):
    """BasicManager.integrate (manager.py:393-405) on pre-drawn float32 w batches (nitn, neval, d).
    The model stays in train mode (batch-statistics BN) as in the reference."""
    cur = dict(params)
    means, variances = [], []
    for w in w_draws:
        upd = {}
        XJ = model_forward(np.asarray(w, dtype=np.float64), cur, steps, train, upd)
        cur.update(upd)
        fres = f(XJ[:, :-1]) * XJ[:, -1]
        _, m, v, _ = moments(fres)
        means.append(m)
        variances.append(v)
    sig, sig_err = integrate_combine(means, variances, w_draws[0].shape[0])
    return sig, sig_err, np.array(means), np.array(variances), cur
