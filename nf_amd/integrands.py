"""Integrands known to libnis_hip.so.  Each object is an ordinary callable on a (N, d) tensor (torch
ops, any device -- this is what a user of the reference would write, README.md:35-36) and ALSO carries
`nis_kind` / `nis_params`, which lets `BasicManager.integrate` evaluate f(X)*J inside the HIP kernel
without materialising X."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _hip_eval(f, x):
    """f(x) in one HIP launch (`nis_integrand_eval`) when x is a float64 CUDA matrix that autograd does not track --
    the case of the training loop's `f(XJ[:, :-1].detach())` (manager.py:241-246); None otherwise (the torch
    expression below then serves: CPU tensors, other dtypes, differentiable inputs)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.double and x.dim() == 2 and not x.requires_grad
            and x.stride(1) == 1 and x.shape[0] > 0):
        return None
    L = _lib.lib()
    out = torch.empty(x.shape[0], dtype=torch.double, device=x.device)
    p = list(f.nis_params) + [0.0] * (4 - len(f.nis_params))
    with _lib.on_device(x.device):
        rc = L.nis_integrand_eval(f.nis_kind, (C.c_double * 4)(*p), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1],
                                  out.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "nis_integrand_eval")
    return out


class Camel:
    """exp(-|x-a|^2/s^2) + exp(-|x-b|^2/s^2) on the first two columns (README.md:35-36: a=.75, b=.25, s=.2)."""
    nis_kind = _lib.NIS_F_CAMEL

    def __init__(self, a=0.75, b=0.25, s=0.2):
        self.a, self.b, self.s = a, b, s
        self.nis_params = (a, b, s)

    def __call__(self, x):
        y = _hip_eval(self, x)
        if y is not None:
            return y
        a, b, s = self.a, self.b, self.s
        return torch.exp(-((x[:, 0] - a) ** 2 + (x[:, 1] - a) ** 2) / (s ** 2)) + \
            torch.exp(-((x[:, 0] - b) ** 2 + (x[:, 1] - b) ** 2) / (s ** 2))


class Normal:
    """manager.py:39-40 `normal(x, mu, sigma, n_flow)`."""
    nis_kind = _lib.NIS_F_NORMAL

    def __init__(self, mu, sigma, n_flow):
        self.mu, self.sigma, self.n_flow = mu, sigma, n_flow
        self.nis_params = (mu, sigma)

    def __call__(self, x):
        y = _hip_eval(self, x)
        if y is not None:
            return y
        return (torch.exp(-torch.sum((x - self.mu) ** 2 / (2 * self.sigma ** 2), -1))) / \
            (self.sigma * np.sqrt((2 * np.pi) ** self.n_flow))


class One:
    nis_kind = _lib.NIS_F_ONE
    nis_params = ()

    def __call__(self, x):
        return torch.ones_like(x[:, 0])


class RamboWeight:
    """Phase-space integrand: f(x) = weight of `FlatInvertiblePhasespace.generateKinematics_batch(E_cm, x, cuts)` (unit
    matrix element), i.e. how the reference's experiment driver feeds RAMBO to the flow.
    Callable on a (N, d) CUDA tensor like any user integrand; `from_planes` lets `integrate` evaluate it straight from the
    flow's column planes (no (N, d) matrix, no momenta written)."""

    def __init__(self, generator, E_cm, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1, pdgs=(0, 0)):
        self.gen, self.E_cm = generator, E_cm
        self.cuts = (pT_mincut, delR_mincut, rap_maxcut)
        self.pdgs = list(pdgs)

    def __call__(self, x):
        _, wgt = self.gen.generateKinematics_batch(self.E_cm, x, self.cuts[0], self.cuts[1], self.cuts[2], self.pdgs)
        return wgt

    def from_planes(self, planes, out_col, N):
        wgt, status = self.gen.weights_from_planes(self.E_cm, planes, out_col, N, *self.cuts, pdgs=self.pdgs)
        return wgt

    def moments_from_planes(self, planes, out_col, jplane, N, out4, status):
        """(G, d+1, N) kept planes -> (G, 4) moment records of weight * J, one launch set, no host sync."""
        self.gen.moments_from_planes(self.E_cm, planes, out_col, jplane, N, out4, status, *self.cuts, pdgs=self.pdgs)

    def check_status(self, status):
        return self.gen.check_status(status)
