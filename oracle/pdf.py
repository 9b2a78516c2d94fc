"""Parton-density interpolation as LHAPDF 6 does it (oracle, test infrastructure only) -- PARITY UNPINNED.

What the reference calls: `pdf.xfxQ2(pdg, x, scale2) / x` with an `lhapdf.PDF` object
(/root/reference/nisrep/PhaseSpace/flat_phase_space_generator.py:120-137, scale fixed at 91.188**2, :183-187).  LHAPDF is a
third-party C++ library (README.md:23 pins 6.3.0) that is NOT vendored in /root/reference and not installed here, and no
PDF grid file ships with the reference, so neither the library nor a golden vector exists to pin this file against: it
restates the PUBLISHED algorithm (Buckley et al., "LHAPDF6: parton density access in the LHC precision era", EPJC 75 (2015)
132, section 3.2; LHAPDF 6.3 `LogBicubicInterpolator`): a cubic Hermite spline in log x and in log Q^2 whose knot
derivatives are the average of the two adjacent finite differences (one-sided at the grid edges), evaluated first along
log x at the Q^2 knots that bracket the point, then along log Q^2.  The lhagrid1 text format is parsed as documented there
(one or more Q^2 subgrids: x knots, Q knots, parton ids, then nx * nq lines of one value per parton; blocks end with '---').
Outside the grid the argument is frozen at the boundary (LHAPDF's default extrapolator continues instead; the phase-space
map only asks inside: x in [1e-4, 1], Q^2 = 91.188^2).
"""
import numpy as np


class Subgrid:
    def __init__(self, xs, q2s, pids, xf):
        self.xs = np.asarray(xs, dtype=np.float64)
        self.q2s = np.asarray(q2s, dtype=np.float64)
        self.pids = [int(p) for p in pids]
        self.xf = np.asarray(xf, dtype=np.float64).reshape(len(self.xs), len(self.q2s), len(self.pids))
        self.logxs = np.log(self.xs)
        self.logq2s = np.log(self.q2s)


def parse_lhagrid1(text):
    """lhagrid1 text -> list of Subgrid (the YAML-ish header before the first '---' is skipped)."""
    blocks = [b.strip() for b in text.split("---")]
    out = []
    for b in blocks[1:]:
        if not b:
            continue
        lines = [ln for ln in b.splitlines() if ln.strip()]
        xs = [float(v) for v in lines[0].split()]
        qs = [float(v) for v in lines[1].split()]
        pids = [int(v) for v in lines[2].split()]
        vals = np.array([[float(v) for v in ln.split()] for ln in lines[3:]], dtype=np.float64)
        assert vals.shape == (len(xs) * len(qs), len(pids)), (vals.shape, len(xs), len(qs), len(pids))
        out.append(Subgrid(xs, np.square(qs), pids, vals.reshape(len(xs), len(qs), len(pids))))
    return out


def write_lhagrid1(subgrids, header="PdfType: central\nFormat: lhagrid1\n"):
    parts = [header.rstrip("\n")]
    for g in subgrids:
        rows = [" ".join(f"{v:.17e}" for v in g.xs), " ".join(f"{v:.17e}" for v in np.sqrt(g.q2s)), " ".join(str(p) for p in g.pids)]
        rows += [" ".join(f"{v:.17e}" for v in g.xf[i, j]) for i in range(len(g.xs)) for j in range(len(g.q2s))]
        parts.append("\n".join(rows))
    return "\n---\n".join(parts) + "\n---\n"


def _cubic(t, vl, vdl, vh, vdh):
    t2 = t * t
    t3 = t2 * t
    return (2 * t3 - 3 * t2 + 1) * vl + (t3 - 2 * t2 + t) * vdl + (-2 * t3 + 3 * t2) * vh + (t3 - t2) * vdh


def _ddx(f, logxs, ix):
    """d f / d log x at knot ix (array of knots): average of the adjacent differences, one-sided at the ends."""
    n = len(logxs)
    out = np.empty(ix.shape)
    lo, hi = ix == 0, ix == n - 1
    mid = ~(lo | hi)
    out[lo] = (f[1] - f[0]) / (logxs[1] - logxs[0])
    out[hi] = (f[n - 1] - f[n - 2]) / (logxs[n - 1] - logxs[n - 2])
    i = ix[mid]
    out[mid] = 0.5 * ((f[i] - f[i - 1]) / (logxs[i] - logxs[i - 1]) + (f[i + 1] - f[i]) / (logxs[i + 1] - logxs[i]))
    return out


def _interp_x(g, fl, iq, ix, tlogx, dlogx):
    f = g.xf[:, iq, fl]
    return _cubic(tlogx, f[ix], _ddx(f, g.logxs, ix) * dlogx, f[ix + 1], _ddx(f, g.logxs, ix + 1) * dlogx)


def xfx_q2(subgrids, pid, x, q2):
    """x f(x, Q^2) of parton `pid` for arrays x, q2 (same shape or scalar q2)."""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    q2 = np.broadcast_to(np.asarray(q2, dtype=np.float64), x.shape)
    out = np.empty(x.shape)
    kidx = np.zeros(x.shape, dtype=np.int64)                       # subgrid of a point: the last one whose Q^2 range starts at or below it
    for k, g in enumerate(subgrids):
        kidx[q2 >= g.q2s[0]] = k
    for k, g in enumerate(subgrids):
        sel = kidx == k
        if not sel.any():
            continue
        fl = g.pids.index(pid)
        xs = np.clip(x[sel], g.xs[0], g.xs[-1])
        qs = np.clip(q2[sel], g.q2s[0], g.q2s[-1])
        logx, logq = np.log(xs), np.log(qs)
        ix = np.clip(np.searchsorted(g.logxs, logx, side="right") - 1, 0, len(g.xs) - 2)
        iq = np.clip(np.searchsorted(g.logq2s, logq, side="right") - 1, 0, len(g.q2s) - 2)
        dlogx = g.logxs[ix + 1] - g.logxs[ix]
        tlogx = (logx - g.logxs[ix]) / dlogx
        dq1 = g.logq2s[iq + 1] - g.logq2s[iq]
        tlogq = (logq - g.logq2s[iq]) / dq1
        res = np.empty(xs.shape)
        nq = len(g.q2s)
        for iqv in np.unique(iq):
            m = iq == iqv
            a = (ix[m], tlogx[m], dlogx[m])
            vl = _interp_x(g, fl, iqv, *a)
            vh = _interp_x(g, fl, iqv + 1, *a)
            d1 = (vh - vl) / dq1[m]
            if iqv == 0:
                vdl = d1
            else:
                vll = _interp_x(g, fl, iqv - 1, *a)
                vdl = 0.5 * (d1 + (vl - vll) / (g.logq2s[iqv] - g.logq2s[iqv - 1]))
            if iqv + 1 == nq - 1:
                vdh = d1
            else:
                vhh = _interp_x(g, fl, iqv + 2, *a)
                vdh = 0.5 * (d1 + (vhh - vh) / (g.logq2s[iqv + 2] - g.logq2s[iqv + 1]))
            res[m] = _cubic(tlogq[m], vl, vdl * dq1[m], vh, vdh * dq1[m])
        out[sel] = res
    return out


def density(subgrids, pdg, x, q2):
    """get_pdfQ2 of the reference (:120-137): xf / x for gluon and quarks, 1 for every other particle id."""
    if pdg != 21 and abs(pdg) not in range(1, 7):
        return np.ones_like(np.asarray(x, dtype=np.float64))
    return xfx_q2(subgrids, pdg, x, q2) / np.asarray(x, dtype=np.float64)
