"""`FlowSequential`: the torch.nn.Sequential the managers build (same child names as the reference:
"<i>", "roll<i>", "mask<i>", "demask<i>", manager.py:487-492, :547-585) whose forward runs as ONE
libnis_hip flow: permutation layers are folded into plane indices, cells run as HIP kernels."""
import torch

from ..flow import FlowEngine
from ._autograd import run_flow


class FlowSequential(torch.nn.Sequential):
    def __init__(self, *args):
        super().__init__(*args)
        self._engine = None
        self._engine_key = None

    # ---- topology -> (cells, in_col per cell, out_col)
    def plan(self):
        mods = list(self.children())
        cells = [m for m in mods if hasattr(m, "KIND")]
        if not cells:
            raise AttributeError("No coupling cell in the model")
        d = cells[0].flow_size
        cur = list(range(d))
        in_cols = []
        for m in mods:
            if hasattr(m, "KIND"):
                if m.flow_size != d:
                    raise ValueError("all cells of a model must share flow_size")
                in_cols.append(list(cur))
            elif hasattr(m, "remap"):
                cur = m.remap(cur)
            else:
                raise TypeError(f"layer {type(m).__name__} cannot be folded into the HIP flow")
        return cells, in_cols, cur, d

    def engine(self):
        key = tuple(id(m) for m in self.children())
        if self._engine is None or key != self._engine_key:
            cells, in_cols, out_col, d = self.plan()
            self._engine = FlowEngine(cells, in_cols, out_col, d)
            self._engine_key = key
        return self._engine

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_engine_key") else copy.deepcopy(v, memo)
        return new

    def forward(self, x):
        """x = format_input(w): (N, d+1) with the Jacobian column last -> XJ (N, d+1)."""
        return run_flow(self.engine(), list(self.children()), x, has_jac=True, train=self.training)

    def forward_raw(self, w):
        """XJ for raw samples w (N, d) with the unit Jacobian of AddJacobian implied (layers.py:75-77): same result as
        `self(format_input(w))` without assembling the (N, d+1) input; differentiable like `forward`."""
        return run_flow(self.engine(), list(self.children()), w, has_jac=False, train=self.training)

    def flow_forward(self, w, want_xj=False, want_bins=False):
        """No-grad fast path on raw samples w (N,d): skips the AddJacobian concat, leaves the result in
        column planes.  Returns the engine's dict (planes / xj / bins / status)."""
        with torch.no_grad():
            return self.engine().forward(w, False, self.training, want_xj=want_xj, want_bins=want_bins)
