"""Coupling cells with the reference's class names, constructor arguments, attributes and
`state_dict` layout (/root/reference/nisrep/normalizing_flows/layers/coupling_cells.py:73-254),
whose forward pass runs in the HIP kernels of libnis_hip.so.

`cell.NN` is still a torch.nn.Sequential of BatchNorm1d / Linear / ReLU / Reshape modules, built in the
reference's order (same RNG consumption at construction, same parameter names
"NN.<idx>.<weight|bias|running_mean|running_var|num_batches_tracked>"), but it is only a parameter
container: the modules are never called on the product path.
"""
import torch

from .layers import Reshape
from ... import _lib
from ...flow import FlowEngine
from .._autograd import run_flow


def _conditioner(pass_through_size, sizes, reshape, hidden_bias=False):
    """BN(P) -> Linear(no bias) -> BN -> ReLU -> [Linear(no bias) -> BN -> ReLU]* -> Linear(+bias) -> Reshape
    (coupling_cells.py:236-254; the same stack is written inline in PWLin, :84-104).  AffineCoupling builds the
    same stack with biased hidden Linears (coupling_cells.py:24-40): hidden_bias=True."""
    mods = [torch.nn.BatchNorm1d(pass_through_size),
            torch.nn.Linear(pass_through_size, sizes[0], bias=hidden_bias),
            torch.nn.BatchNorm1d(sizes[0]),
            torch.nn.ReLU()]
    old = sizes[0]
    for size in sizes[1:-1]:
        mods += [torch.nn.Linear(old, size, bias=hidden_bias), torch.nn.BatchNorm1d(size), torch.nn.ReLU()]
        old = size
    mods.append(torch.nn.Linear(old, sizes[-1]))
    mods.append(Reshape(reshape[0], reshape[1]))
    return torch.nn.Sequential(*mods)


class RectNN(torch.nn.Module):
    """Rectangular conditioner network (coupling_cells.py:230-254)."""

    def __init__(self, pass_through_size, sizes, reshape):
        super().__init__()
        self.NN = _conditioner(pass_through_size, sizes, reshape)


class _PiecewiseCell(torch.nn.Module):
    KIND = None
    OUT_PER_BIN = None

    HIDDEN_BIAS = False

    def __init__(self, flow_size, pass_through_size, n_bins, NN_layers):
        super().__init__()
        self.pass_through_size = pass_through_size
        self.flow_size = flow_size
        self.transform_size = flow_size - pass_through_size
        self.n_bins = n_bins
        out = self.outputs_per_column()
        sizes = list(NN_layers) + [self.transform_size * out]
        self.NN = _conditioner(pass_through_size, sizes, self.reshape_args(out), self.HIDDEN_BIAS)
        self.hidden_widths = [int(s) for s in sizes[:-1]] if len(sizes) > 1 else [int(sizes[0])]
        self._flat = None
        self._running = None
        self._tracked = None
        self._engine = None

    def outputs_per_column(self):
        raise NotImplementedError

    def reshape_args(self, out):
        return (self.transform_size, out)

    def _own_engine(self):
        if self._engine is None:
            d = self.flow_size
            self._engine = FlowEngine([self], [list(range(d))], list(range(d)), d)
        return self._engine

    def __deepcopy__(self, memo):
        # the ctypes handle is not copyable: drop it in the copy, it is rebuilt lazily
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_engine":
                new.__dict__[k] = None
            elif k in ("_flat", "_running", "_tracked", "_sd_cache"):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def forward(self, x):
        """x: (N, flow_size+1) = [pass-through | transformed | running Jacobian] -> same shape
        (coupling_cells.py:107-142 / :159-228), differentiable w.r.t. x and the conditioner parameters."""
        return run_flow(self._own_engine(), [self], x, has_jac=True, train=self.training)


class PWLin(_PiecewiseCell):
    """Piecewise-linear coupling cell (coupling_cells.py:73-142)."""
    KIND = _lib.NIS_PWLIN

    def outputs_per_column(self):
        return self.n_bins


class PWQuad(_PiecewiseCell):
    """Piecewise-quadratic coupling cell (coupling_cells.py:144-228)."""
    KIND = _lib.NIS_PWQUAD

    def outputs_per_column(self):
        return 2 * self.n_bins + 1


class AffineCoupling(_PiecewiseCell):
    """Affine coupling cell (coupling_cells.py:6-70): y = atan(20 exp(z0) x + relu(z1)) / (pi/2), conditioner with
    biased hidden Linears and a Reshape(2, T) head.  Same constructor as the reference (no n_bins)."""
    KIND = _lib.NIS_AFFINE
    HIDDEN_BIAS = True

    def __init__(self, flow_size, pass_through_size, NN_layers):
        super().__init__(flow_size, pass_through_size, 0, NN_layers)
        del self.n_bins                                       # the reference cell has no such attribute

    def outputs_per_column(self):
        return 2

    def reshape_args(self, out):
        return (2, self.transform_size)
