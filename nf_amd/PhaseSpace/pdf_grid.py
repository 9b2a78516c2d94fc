"""Parton densities for `FlatInvertiblePhasespace(pdf=...)` on the device.

The reference hands `generateKinematics_batch` an `lhapdf.PDF` object and calls `pdf.xfxQ2(pdg, x, scale2) / x` per sample
through a Python list (/root/reference/nisrep/PhaseSpace/flat_phase_space_generator.py:120-137), always at the fixed
scale 91.188**2 (:184).  LHAPDF itself (a third-party C++ library, not vendored, not installed) stays out of scope; what is
built is its interpolation: `PdfGrid` reads the same `lhagrid1` grid files LHAPDF reads (or arrays), and the density is
evaluated INSIDE `k_rambo` (PDF modes) by the published LHAPDF 6 log-bicubic rule (Buckley et al., EPJC 75 (2015) 132;
`LogBicubicInterpolator`): cubic Hermite in log x and log Q^2, knot derivatives = mean of the adjacent differences
(one-sided at the edges).  The interpolation is linear in the grid values, so for a FIXED scale the log Q^2 direction is
folded into a per-parton table over the x knots once on the host (O(nx) NumPy, like packing parameters), and the kernel
evaluates a 1-D spline in log x: a binary search over the knots, 4 table values, ~60 instructions per beam.
No LHAPDF fixture exists for this path: parity unpinned (checked against oracle/pdf.py, a NumPy restatement of the same
published algorithm, on synthetic grids).  Outside the grid the argument is frozen at the boundary.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class _Sub:
    def __init__(self, xs, q2s, pids, xf):
        self.xs = np.asarray(xs, dtype=np.float64)
        self.q2s = np.asarray(q2s, dtype=np.float64)
        self.pids = [int(p) for p in pids]
        self.xf = np.asarray(xf, dtype=np.float64).reshape(len(self.xs), len(self.q2s), len(self.pids))
        if len(self.xs) < 2 or len(self.q2s) < 2:
            raise ValueError("a PDF subgrid needs at least 2 knots in x and in Q")
        self.logq2s = np.log(self.q2s)


class Pdf1d(C.Structure):                                 # nis_pdf1d
    _fields_ = [("logx", C.c_void_p), ("xf", C.c_void_p), ("nx", C.c_int32), ("pad", C.c_int32)]


class PdfGrid:
    """One PDF member: lhagrid1 subgrids (ascending in Q^2).  `xfxQ2(pid, x, q2)` mirrors lhapdf.PDF.xfxQ2 for a scalar q2."""

    def __init__(self, subgrids):
        self.subgrids = list(subgrids)
        self._cache = {}

    @classmethod
    def from_arrays(cls, xs, q2s, pids, xf):
        return cls([_Sub(xs, q2s, pids, xf)])

    @classmethod
    def from_lhagrid1(cls, path_or_text):
        text = path_or_text if "\n" in path_or_text else open(path_or_text).read()
        subs = []
        for b in [b.strip() for b in text.split("---")][1:]:
            if not b:
                continue
            lines = [ln for ln in b.splitlines() if ln.strip()]
            xs = [float(v) for v in lines[0].split()]
            qs = [float(v) for v in lines[1].split()]
            pids = [int(v) for v in lines[2].split()]
            vals = np.array([[float(v) for v in ln.split()] for ln in lines[3:]], dtype=np.float64)
            if vals.shape != (len(xs) * len(qs), len(pids)):
                raise ValueError(f"lhagrid1 block: {vals.shape} values for {len(xs)} x {len(qs)} knots and {len(pids)} partons")
            subs.append(_Sub(xs, np.square(qs), pids, vals.reshape(len(xs), len(qs), len(pids))))
        if not subs:
            raise ValueError("no grid block found (lhagrid1 blocks are separated by '---')")
        return cls(subs)

    def has(self, pid):
        return any(pid in g.pids for g in self.subgrids)

    def collapse(self, pid, q2):
        """(log x knots, x f at Q^2 = q2 on the knots): the log Q^2 cubic of the LHAPDF rule applied per x knot."""
        q2 = float(q2)
        g = self.subgrids[0]
        for cand in self.subgrids:
            if q2 >= cand.q2s[0]:
                g = cand
        fl = g.pids.index(pid)
        f = g.xf[:, :, fl]
        lq = float(np.log(min(max(q2, g.q2s[0]), g.q2s[-1])))
        nq = len(g.q2s)
        iq = int(min(max(np.searchsorted(g.logq2s, lq, side="right") - 1, 0), nq - 2))
        d1 = g.logq2s[iq + 1] - g.logq2s[iq]
        t = (lq - g.logq2s[iq]) / d1
        vl, vh = f[:, iq], f[:, iq + 1]
        s1 = (vh - vl) / d1
        vdl = s1 if iq == 0 else 0.5 * (s1 + (vl - f[:, iq - 1]) / (g.logq2s[iq] - g.logq2s[iq - 1]))
        vdh = s1 if iq + 1 == nq - 1 else 0.5 * (s1 + (f[:, iq + 2] - vh) / (g.logq2s[iq + 2] - g.logq2s[iq + 1]))
        t2, t3 = t * t, t * t * t
        val = (2 * t3 - 3 * t2 + 1) * vl + (t3 - 2 * t2 + t) * (vdl * d1) + (-2 * t3 + 3 * t2) * vh + (t3 - t2) * (vdh * d1)
        return np.log(g.xs), val

    def table(self, pid, q2, dev):
        """Device copy of `collapse(pid, q2)` as a `Pdf1d` descriptor (cached per parton, scale and device)."""
        key = (int(pid), float(q2), str(torch.device(dev)))
        if key not in self._cache:
            lx, val = self.collapse(pid, q2)
            self._cache[key] = (torch.as_tensor(lx, dtype=torch.double).to(dev), torch.as_tensor(val, dtype=torch.double).to(dev))
        lx, val = self._cache[key]
        return Pdf1d(lx.data_ptr(), val.data_ptr(), lx.numel(), 0)

    def xfxQ2(self, pid, x, q2):
        """x f(x, Q^2) for a CUDA tensor x and a scalar Q^2 (lhapdf.PDF.xfxQ2's argument order; the reference passes a
        constant scale, flat_phase_space_generator.py:184)."""
        _lib.require_gpu()
        if torch.is_tensor(q2):
            if q2.numel() > 1 and not bool((q2 == q2.reshape(-1)[0]).all()):
                raise NotImplementedError("per-sample scales: the phase-space map uses one fixed scale (reference :184)")
            q2 = float(q2.reshape(-1)[0])
        if not torch.is_tensor(x) or not x.is_cuda:
            raise _lib.NisError("PdfGrid.xfxQ2 needs a CUDA tensor (no CPU fallback)")
        xx = x.detach().to(torch.double).contiguous()
        out = torch.empty_like(xx)
        tab = self.table(pid, q2, xx.device)
        with _lib.on_device(xx.device):
            rc = _lib.lib().nis_pdf_xfx(C.byref(tab), xx.data_ptr(), xx.numel(), out.data_ptr(), _lib.stream_ptr(xx.device))
        _lib.check(rc, "nis_pdf_xfx")
        return out
