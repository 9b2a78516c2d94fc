"""Pin the NumPy oracle (oracle/) to the golden vectors the reference produced
(tests/golden/*.npz, made by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden
from oracle import flow as oflow
from oracle import rambo as orambo

RTOL = 1e-11   # fp64: different summation order / libm exp only


def _cases(fname):
    return golden(fname).cases()


def plan_from_meta(m):
    if m["kind"] == "pwquad":
        return oflow.plan_pwquad(m["d"], m["n_cells"], m["K"])
    return oflow.plan_pwlin(m["d"], m["P"], m["n_cells"], m["K"], m["roll"])


@pytest.mark.parametrize("case", _cases("cells.npz"))
def test_cell_matches_reference(case, g_cells):
    m = g_cells.meta[case]
    params = {"c." + k: v for k, v in g_cells.sub(case, "sd").items()}
    x = g_cells.get(case, "x")
    want, upd = {}, {}
    y = oflow.cell_forward(x, params, "c", m["kind"], m["P"], m["K"], m["mode"] == "train", upd, want)
    np.testing.assert_array_equal(want["bins"].reshape(-1), g_cells.get(case, "bins").reshape(-1))
    np.testing.assert_allclose(y, g_cells.get(case, "y"), rtol=RTOL, atol=1e-14)
    if g_cells.has(case, "Z"):
        np.testing.assert_allclose(want["Z"], g_cells.get(case, "Z"), rtol=1e-10, atol=1e-12)
    after = g_cells.sub(case, "sd_after")
    for k, v in after.items():
        if m["mode"] == "train":
            np.testing.assert_allclose(upd["c." + k], v, rtol=1e-12, atol=1e-15)
        else:
            np.testing.assert_array_equal(params["c." + k], v)


@pytest.mark.parametrize("case", _cases("models.npz"))
def test_model_topology_and_forward(case, g_models):
    m = g_models.meta[case]
    steps = plan_from_meta(m)
    # children names / pass-through sizes as the reference's Sequential reports them
    names, passes = [], []
    ci = 0
    for st in steps:
        if st[0] == "cell":
            names.append(f"{st[1]}:{'PWQuad' if st[2] == 'pwquad' else 'PWLin'}")
            passes.append(st[3])
            ci = int(st[1])
        elif st[0] == "roll":
            names.append(("roll:RollLayer" if m["kind"] == "pwlin" else f"roll{ci}:RollLayer"))
        elif st[0] == "mask":
            names.append(f"mask{ci + 1 if names else 0}:MaskLayer")
        elif st[0] == "demask":
            names.append(f"demask{ci}:DeMaskLayer")
    assert passes == m["passes"]
    assert [n.split(":")[1] for n in names] == [n.split(":")[1] for n in m["children"]]
    params = g_models.sub(case, "sd")
    w = g_models.get(case, "w")
    upd = {}
    XJ = oflow.model_forward(w, params, steps, True, upd)
    np.testing.assert_allclose(XJ, g_models.get(case, "XJ_train"), rtol=1e-10, atol=1e-13)
    for k, v in g_models.sub(case, "sd_after").items():
        np.testing.assert_allclose(upd[k], v, rtol=1e-11, atol=1e-14)
    params2 = dict(params)
    params2.update(upd)
    XJe = oflow.model_forward(w, params2, steps, False)
    np.testing.assert_allclose(XJe, g_models.get(case, "XJ_eval"), rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("case", _cases("integrate.npz"))
def test_integrate_matches_reference(case, g_integrate):
    m = g_integrate.meta[case]
    steps = plan_from_meta(m)
    params = g_integrate.sub(case, "sd")
    f = oflow.camel if case.startswith("camel") else (lambda x: oflow.normal(x, 0.5, 0.25, 8))
    sig, sig_err, means, variances, after = oflow.integrate(f, params, steps, g_integrate.get(case, "w"))
    np.testing.assert_allclose(means, g_integrate.get(case, "mean"), rtol=1e-10)
    np.testing.assert_allclose(variances, g_integrate.get(case, "var"), rtol=1e-9)
    np.testing.assert_allclose(sig, g_integrate.get(case, "sig"), rtol=2e-6)       # float32 results
    np.testing.assert_allclose(sig_err, g_integrate.get(case, "sig_err"), rtol=2e-6)
    for k, v in g_integrate.sub(case, "sd_after").items():
        np.testing.assert_allclose(after[k], v, rtol=1e-10, atol=1e-14)


def test_torch_restatement_matches_reference_and_numpy(g_integrate):
    """oracle/flow_torch.py (bench.py's all-core CPU baseline) against the reference's own integrate() vectors and the NumPy oracle."""
    import torch
    from oracle import flow_torch as otorch
    case = [c for c in _cases("integrate.npz") if c.startswith("camel")][0]
    m = g_integrate.meta[case]
    steps = plan_from_meta(m)
    params = g_integrate.sub(case, "sd")
    w = g_integrate.get(case, "w")
    sig, sig_err, means, variances, after = otorch.integrate(otorch.camel, otorch.to_torch(params), steps,
                                                             [torch.as_tensor(x) for x in w])
    np.testing.assert_allclose(means.numpy(), g_integrate.get(case, "mean"), rtol=2e-6)        # float32 stores, as the reference
    np.testing.assert_allclose(variances.numpy(), g_integrate.get(case, "var"), rtol=2e-6)
    np.testing.assert_allclose(sig, g_integrate.get(case, "sig"), rtol=2e-6)
    np.testing.assert_allclose(sig_err, g_integrate.get(case, "sig_err"), rtol=2e-6)
    for k, v in g_integrate.sub(case, "sd_after").items():
        np.testing.assert_allclose(after[k].numpy(), v, rtol=1e-10, atol=1e-14)
    rng = np.random.default_rng(3)
    x = rng.random((777, 2))
    a = oflow.model_forward(x, params, steps, True)
    b = otorch.model_forward(torch.as_tensor(x), otorch.to_torch(params), steps, True).numpy()
    np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("case", _cases("rambo.npz"))
def test_rambo_matches_reference(case, g_rambo):
    m = g_rambo.meta[case]
    want = {}
    mom, wgt = orambo.generate_kinematics(m["initial"], m["final"], m["E_cm"], g_rambo.get(case, "r"),
                                          m["cuts"][0], m["cuts"][1], m["cuts"][2],
                                          pdf_active=m["pdf_active"], tau=m["tau"], want=want)
    if g_rambo.has(case, "u"):
        np.testing.assert_allclose(want["u"], g_rambo.get(case, "u"), rtol=1e-12, atol=1e-15)
    scale = np.abs(g_rambo.get(case, "momenta")).max()
    np.testing.assert_allclose(mom, g_rambo.get(case, "momenta"), rtol=1e-10, atol=1e-11 * scale)
    np.testing.assert_allclose(wgt, g_rambo.get(case, "weight"), rtol=1e-10)
    # the cut mask must agree exactly
    np.testing.assert_array_equal(wgt == 0, g_rambo.get(case, "weight") == 0)


@pytest.mark.parametrize("case", _cases("rambo.npz"))
def test_rambo_torch_restatement_matches_reference(case, g_rambo):
    """oracle/rambo_torch.py (the all-core CPU baseline of bench.py for the 2 -> n map) against the reference's golden vectors, at the
    tolerances the NumPy oracle holds; and bit-for-bit the same cut mask."""
    import torch
    from oracle import rambo_torch as otr
    m = g_rambo.meta[case]
    mom, wgt = otr.generate_kinematics(m["initial"], m["final"], m["E_cm"], torch.from_numpy(np.ascontiguousarray(g_rambo.get(case, "r"))),
                                       m["cuts"][0], m["cuts"][1], m["cuts"][2], pdf_active=m["pdf_active"], tau=m["tau"])
    mom, wgt = mom.numpy(), wgt.numpy()
    scale = np.abs(g_rambo.get(case, "momenta")).max()
    np.testing.assert_allclose(mom, g_rambo.get(case, "momenta"), rtol=1e-10, atol=1e-11 * scale)
    np.testing.assert_allclose(wgt, g_rambo.get(case, "weight"), rtol=1e-10)
    np.testing.assert_array_equal(wgt == 0, g_rambo.get(case, "weight") == 0)


def test_rambo_invariants(g_rambo):
    """Known answers independent of the reference: momentum conservation, on-shell legs,
    flat massless weight."""
    for case in g_rambo.cases():
        m = g_rambo.meta[case]
        mom, wgt = orambo.generate_kinematics(m["initial"], m["final"], m["E_cm"], g_rambo.get(case, "r"),
                                              pdf_active=m["pdf_active"], tau=m["tau"])
        tot_in = mom[:, :2].sum(1)
        tot_out = mom[:, 2:].sum(1)
        np.testing.assert_allclose(tot_out, tot_in, atol=1e-9 * m["E_cm"])
        msq = mom[:, 2:, 0] ** 2 - (mom[:, 2:, 1:] ** 2).sum(-1)
        np.testing.assert_allclose(msq, np.broadcast_to(np.asarray(m["final"])[None, :] ** 2, msq.shape),
                                   atol=1e-8 * m["E_cm"] ** 2)
    m = g_rambo.meta["m2to2_massless"]
    _, wgt = orambo.generate_kinematics(m["initial"], m["final"], m["E_cm"], g_rambo.get("m2to2_massless", "r"))
    np.testing.assert_allclose(wgt, orambo.flat_weight(m["E_cm"], 2) / (2 * m["E_cm"] ** 2), rtol=1e-13)


def test_camel_known_integral():
    """Truncated double Gaussian over [0,1]^2: 0.2323 by quadrature."""
    from math import erf, sqrt, pi
    s = 0.2
    g = lambda c: 0.5 * sqrt(pi) * s * (erf((1 - c) / s) - erf((0 - c) / s))
    truth = g(0.75) ** 2 + g(0.25) ** 2
    rng = np.random.default_rng(0)
    x = rng.random((400000, 2))
    est = oflow.camel(x).mean()
    assert abs(est - truth) < 5e-3 * truth * 3


# --------------------------------------------------------------------------- AffineCoupling
@pytest.mark.parametrize("case", [c for c in _cases("affine.npz") if "model" not in c])
def test_affine_cell_matches_reference(case, g_affine):
    m = g_affine.meta[case]
    params = {"c." + k: v for k, v in g_affine.sub(case, "sd").items()}
    x = g_affine.get(case, "x")
    want, upd = {}, {}
    y = oflow.cell_forward(x, params, "c", "affine", m["P"], 0, m["mode"] == "train", upd, want)
    np.testing.assert_allclose(y, g_affine.get(case, "y"), rtol=RTOL, atol=1e-14)
    if g_affine.has(case, "Z"):
        np.testing.assert_allclose(want["Z"], g_affine.get(case, "Z"), rtol=1e-10, atol=1e-12)
    for k, v in g_affine.sub(case, "sd_after").items():
        if m["mode"] == "train":
            np.testing.assert_allclose(upd["c." + k], v, rtol=1e-12, atol=1e-15)
        else:
            np.testing.assert_array_equal(params["c." + k], v)


@pytest.mark.parametrize("case", [c for c in _cases("affine.npz") if "model" in c])
def test_affine_model_topology_and_forward(case, g_affine):
    m = g_affine.meta[case]
    steps = oflow.plan_affine(m["d"], m["P"], m["n_cells"], m["roll"])
    names = [f"{st[1]}:AffineCoupling" if st[0] == "cell" else "roll:RollLayer" for st in steps]
    assert names == m["children"] and [st[3] for st in steps if st[0] == "cell"] == m["passes"]
    params = g_affine.sub(case, "sd")
    w = g_affine.get(case, "w")
    upd = {}
    XJ = oflow.model_forward(w, params, steps, True, upd)
    np.testing.assert_allclose(XJ, g_affine.get(case, "XJ_train"), rtol=1e-10, atol=1e-13)
    for k, v in g_affine.sub(case, "sd_after").items():
        np.testing.assert_allclose(upd[k], v, rtol=1e-11, atol=1e-14)
    params2 = dict(params)
    params2.update(upd)
    XJe = oflow.model_forward(w, params2, steps, False)
    np.testing.assert_allclose(XJe, g_affine.get(case, "XJ_eval"), rtol=1e-10, atol=1e-13)


# ------------------------------------------------------------------------------ PDF grids (parity unpinned: no LHAPDF fixture exists)
def synthetic_pdf_grid(two_subgrids=False):
    """A smooth toy parton set on lhagrid1-like knots: x f = x^(-0.2 + 0.01 p) (1 - x)^(3 + 0.1 |p|) (1 + 0.05 ln Q^2)."""
    from oracle import pdf as opdf
    xs = np.concatenate([np.logspace(-6, -1, 40, endpoint=False), np.linspace(0.1, 1.0, 30)])
    pids = [-2, -1, 1, 2, 21]

    def f(x, q2, p):
        return (x ** (-0.2 + 0.01 * p)) * ((1 - x) ** (3 + 0.1 * abs(p))) * (1 + 0.05 * np.log(q2))

    def sub(q2s):
        return opdf.Subgrid(xs, q2s, pids, np.array([[[f(x, q, p) for p in pids] for q in q2s] for x in xs]))

    if two_subgrids:
        return [sub(np.logspace(0.5, 2.0, 6)), sub(np.logspace(2.0, 8.0, 20))], f
    return [sub(np.logspace(0.5, 8.0, 25))], f


def test_pdf_oracle_and_host_collapse():
    """oracle/pdf.py (the published LHAPDF 6 log-bicubic rule): lhagrid1 round trip, exact at the knots, close to the smooth
    function between them; the product's host-side fold of the log Q^2 direction (PdfGrid.collapse) gives the oracle's
    values on the x knots."""
    from oracle import pdf as opdf
    from nf_amd.PhaseSpace.pdf_grid import PdfGrid
    for two in (False, True):
        subs, f = synthetic_pdf_grid(two)
        back = opdf.parse_lhagrid1(opdf.write_lhagrid1(subs))
        assert len(back) == len(subs)
        for a, b in zip(subs, back):
            np.testing.assert_array_equal(a.xf, b.xf)
            np.testing.assert_allclose(a.q2s, b.q2s, rtol=1e-15)
        g = subs[-1]
        np.testing.assert_allclose(opdf.xfx_q2(subs, 21, g.xs[3:-3], g.q2s[4]), g.xf[3:-3, 4, 4], rtol=1e-13)
        rng = np.random.default_rng(1)
        x, q2 = 10 ** rng.uniform(-4, np.log10(0.8), 3000), 10 ** rng.uniform(1, 7, 3000)
        v = opdf.xfx_q2(subs, 2, x, q2)
        assert np.max(np.abs(v - f(x, q2, 2)) / f(x, q2, 2)) < 2e-3
        grid = PdfGrid.from_lhagrid1(opdf.write_lhagrid1(subs))
        for q in (91.188 ** 2, 10.0, 3.5e7, g.q2s[5]):
            lx, val = grid.collapse(-1, q)
            want = opdf.xfx_q2(subs, -1, np.exp(lx), q)
            np.testing.assert_allclose(val, want, rtol=1e-12)
    assert float(opdf.density(subs, 22, np.array([0.3]), 8000.)[0]) == 1.0          # not a parton: density 1 (reference :128-129)
