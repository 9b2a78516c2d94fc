"""GPU parity tests (-m gpu): the HIP path, called through the drop-in classes / the C ABI, against
(1) the golden vectors the reference produced and (2) the NumPy oracle on fresh seeded inputs, plus
size-independent properties at BASELINE.json's full batch sizes.

Tolerances (float64 everywhere): bin indices bit-exact; values rtol 1e-9 (different summation order in
the conditioner, device libm exp/log within 1-2 ulp, batch statistics merged pairwise instead of
sequentially).  float32 results of integrate(): rtol 2e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from torch_adjoint import model_torch
from oracle import flow as oflow
from oracle import rambo as orambo

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-9, 1e-12


def dev():
    return torch.device("cuda:0")


def t2n(t):
    return t.detach().cpu().numpy()


def make_cell(m, sd):
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    cls = PWQuad if m["kind"] == "pwquad" else PWLin
    cell = cls(m["d"], m["P"], m["K"], list(m["NN"])).double()
    cell.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return cell.to(dev())


def build_manager(m, sd=None):
    from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
    if m["kind"] == "pwquad":
        NF = PWQuadManager(n_flow=m["d"])
        NF.create_model(m["n_cells"], m["K"], list(m["NN"]))
    else:
        NF = PWLinManager(n_flow=m["d"])
        NF.create_model(m["P"], m["n_cells"], m["K"], list(m["NN"]), m["roll"])
    if sd is not None:
        NF._model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return NF


def plan_from_meta(m):
    if m["kind"] == "pwquad":
        return oflow.plan_pwquad(m["d"], m["n_cells"], m["K"])
    return oflow.plan_pwlin(m["d"], m["P"], m["n_cells"], m["K"], m["roll"])


# ------------------------------------------------------------------------------ single cells
@pytest.mark.parametrize("case", golden("cells.npz").cases())
def test_cell_vs_reference_golden(case, g_cells):
    m = g_cells.meta[case]
    cell = make_cell(m, g_cells.sub(case, "sd"))
    cell.train(m["mode"] == "train")
    x = torch.as_tensor(g_cells.get(case, "x")).to(dev())
    res = cell._own_engine().forward(x, True, cell.training, want_xj=True, want_bins=True)
    torch.cuda.synchronize()
    assert int(res["status"][0]) == 0
    bins = t2n(res["bins"]).T.reshape(-1)                       # (T,N) -> reference order (N,T)
    np.testing.assert_array_equal(bins, g_cells.get(case, "bins").reshape(-1))
    np.testing.assert_allclose(t2n(res["xj"]), g_cells.get(case, "y"), rtol=RTOL, atol=ATOL)
    sd_now = cell.state_dict()
    for k, v in g_cells.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd_now[k]), v, rtol=1e-11, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("kind,d,P,K,NN,N", [("pwquad", 2, 1, 4, [3, 3, 3], 5000), ("pwquad", 6, 2, 9, [12, 5], 3001),
                                             ("pwlin", 4, 2, 50, [9], 4097), ("pwquad", 12, 5, 20, [40, 33], 2500),
                                             ("pwlin", 9, 3, 128, [64, 64], 700), ("pwquad", 3, 2, 32, [2, 2, 2, 2], 999),
                                             # matrix-core heads whose LDS image decides the columns per bijector batch (CellK::mf_ncb)
                                             ("pwlin", 5, 2, 128, [32, 32], 600), ("pwquad", 6, 3, 32, [32, 32], 800),
                                             ("pwlin", 8, 4, 64, [32, 32, 32], 1500),
                                             # PWQuad with more than 32 bins: heights / widths in per-thread scratch arrays (pwquad_column_big);
                                             # wide conditioners take the fp64 VALU kernel there (the matrix-core image kernel stops at 32)
                                             ("pwquad", 3, 1, 33, [4, 4], 2000), ("pwquad", 5, 2, 100, [8, 6], 1200),
                                             ("pwquad", 4, 2, 128, [3], 900), ("pwquad", 6, 3, 64, [32, 32], 700),
                                             # matrix-core PWQuad heads with K == KM: the relative last-layer image (NIS_PACK_REL), 2K - 1 rows per
                                             # column -- column batches that start inside a row tile, one column, every K the image exists for
                                             ("pwquad", 7, 3, 4, [16, 16], 1100), ("pwquad", 9, 2, 8, [40, 40], 900),
                                             ("pwquad", 5, 4, 16, [24], 1300), ("pwquad", 11, 2, 16, [64, 64], 650),
                                             # width 16 with seven pass-through columns: 21 co-moments fit the BN record but not the 16 columns of
                                             # the merge -- BN_1 must NOT be derived from them there (found by NIS_TEST_SWEEP_SEED=5: values 1 % off)
                                             ("pwlin", 8, 7, 4, [3, 11], 477), ("pwquad", 11, 7, 16, [16, 2], 385), ("pwquad", 9, 6, 8, [16, 16], 500)])
@pytest.mark.parametrize("train", [True, False])
def test_cell_vs_oracle_seeded(kind, d, P, K, NN, N, train):
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    torch.manual_seed(d * 100 + K)
    cell = (PWQuad if kind == "pwquad" else PWLin)(d, P, K, list(NN)).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.3 * torch.randn_like(p))
        for mod in cell.NN:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(0, 0.5)
                mod.running_var.uniform_(0.1, 1.0)
    params = {"c." + k: t2n(v) for k, v in cell.state_dict().items()}
    x = torch.rand(N, d + 1, dtype=torch.double)
    x[:, -1] += 0.5
    if kind == "pwlin":
        x[:, :d] *= (1 - 1e-12)
    want, upd = {}, {}
    y_ref = oflow.cell_forward(x.numpy(), params, "c", kind, P, K, train, upd, want)
    cell = cell.to(dev()).train(train)
    res = cell._own_engine().forward(x.to(dev()), True, train, want_xj=True, want_bins=True)
    np.testing.assert_array_equal(t2n(res["bins"]).T, want["bins"])
    np.testing.assert_allclose(t2n(res["xj"]), y_ref, rtol=RTOL, atol=ATOL)
    if train:
        sd = cell.state_dict()
        for k, v in upd.items():
            np.testing.assert_allclose(t2n(sd[k[2:]]), v, rtol=1e-11, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("d,P,K,NN", [(3, 1, 8, [4, 4]), (6, 3, 16, [32, 32]), (5, 2, 4, [16, 16]), (4, 2, 20, [24])],
                         ids=["narrow-K8", "matrixcore-K16-relative-image", "matrixcore-K4-relative-image", "matrixcore-K20-reference-rows"])
def test_pwquad_vanishing_width_takes_the_reference_search(d, P, K, NN):
    """A width that vanishes in the cumulative sum (exp(z) 1e-35 of its neighbours) leaves two EQUAL bin edges; the reference's argmax
    (coupling_cells.py:198-202) takes the first of them, so samples right of the pair land in the zero-width bin.  The kernels' fast
    searches (multiplied-out comparisons; in the matrix-core image kernel a COUNT of the edges below x) cannot see that and must hand the
    wave to the search with the reference's own divisions (pwquad_bin_divide): bins bit-identical to the oracle, values wherever the
    selected bin has a width."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWQuad
    torch.manual_seed(K)
    cell = PWQuad(d, P, K, list(NN)).double()
    T, k0 = d - P, K // 2
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.2 * torch.randn_like(p))
        last = [m for m in cell.NN if isinstance(m, torch.nn.Linear)][-1]
        for t in range(T):                                       # width row k0 of every column: exp(-80) relative
            last.weight[t * (2 * K + 1) + K + 1 + k0].zero_()
            last.bias[t * (2 * K + 1) + K + 1 + k0] = -80.0
    params = {"c." + k: t2n(v) for k, v in cell.state_dict().items()}
    N = 4000
    x = torch.rand(N, d + 1, dtype=torch.double)
    x[:, -1] += 0.5
    for train in (True, False):
        want, upd = {}, {}
        y_ref = oflow.cell_forward(x.numpy(), params, "c", "pwquad", P, K, train, upd, want)
        m = PWQuad(d, P, K, list(NN)).double()
        m.load_state_dict(cell.state_dict())
        m = m.to(dev()).train(train)
        res = m._own_engine().forward(x.to(dev()), True, train, want_xj=True, want_bins=True)
        bins = t2n(res["bins"]).T
        np.testing.assert_array_equal(bins, want["bins"])
        assert (bins == k0).any(), "no sample reached the zero-width bin: the case is not exercised"
        ok = (bins != k0).all(axis=1)                            # rows whose every column sits in a bin with a width
        assert ok.sum() > N // 4
        np.testing.assert_allclose(t2n(res["xj"])[ok], y_ref[ok], rtol=RTOL, atol=ATOL)


def test_cell_module_call_is_differentiable():
    """cell(x) is the reference's nn.Module surface: forward values from HIP, gradients available."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWQuad
    torch.manual_seed(3)
    cell = PWQuad(3, 1, 5, [4, 4]).double().to(dev())
    x = torch.rand(64, 4, dtype=torch.double, device=dev(), requires_grad=True)
    y = cell(x)
    (y[:, -1].var() + y[:, 1].sum()).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    g = [p.grad for p in cell.parameters()]
    assert all(gi is not None and torch.isfinite(gi).all() for gi in g)
    # finite-difference check of one weight against the HIP forward
    p = cell.NN[1].weight
    eps = 1e-6
    with torch.no_grad():
        base = cell(x.detach())
        p[0, 0] += eps
        pert = cell(x.detach())
        p[0, 0] -= eps
    fd = ((pert[:, -1].var() + pert[:, 1].sum()) - (base[:, -1].var() + base[:, 1].sum())) / eps
    assert abs(float(fd) - float(p.grad[0, 0])) < 1e-4 * max(1.0, abs(float(fd)))


def test_hip_backward_general_narrow_models(monkeypatch):
    """Adjoint kernels on manager-built models with several transformed columns, masks, width-8 conditioners."""
    from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
    cases = [("pwquad", dict(d=5, n_cells=2, K=6, NN=[8, 8])), ("pwquad", dict(d=7, n_cells=3, K=16, NN=[6])),
             ("pwquad", dict(d=2, n_cells=2, K=32, NN=[3, 3, 3])), ("pwquad", dict(d=4, n_cells=4, K=27, NN=[8])),
             ("pwquad", dict(d=9, n_cells=4, K=5, NN=[7, 4])), ("pwlin", dict(d=3, P=1, n_cells=3, K=33, NN=[5, 5], roll=1)),
             ("pwlin", dict(d=6, P=4, n_cells=2, K=128, NN=[8], roll=2)),
             ("pwquad", dict(d=3, n_cells=2, K=40, NN=[4, 4])), ("pwquad", dict(d=5, n_cells=3, K=128, NN=[8]))]     # > 32 bins: scratch-array head
    for kind, c in cases:
        torch.manual_seed(c["d"])
        if kind == "pwquad":
            NF = PWQuadManager(n_flow=c["d"])
            NF.create_model(c["n_cells"], c["K"], list(c["NN"]))
        else:
            NF = PWLinManager(n_flow=c["d"])
            NF.create_model(c["P"], c["n_cells"], c["K"], list(c["NN"]), c["roll"])
        model = NF._model.train()
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.1 * torch.randn_like(p))
        assert model.engine().backward_supported(), c
        N, d = 2049, c["d"]
        x0 = torch.rand(N, d + 1, dtype=torch.double, device=dev()) * (1 - 1e-9)
        R = torch.randn(N, d + 1, dtype=torch.double, device=dev())

        def grads(force_torch):
            for p in model.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            out = model_torch(model, x) if force_torch else model(x)
            (out * R).sum().backward()
            return x.grad.clone(), [p.grad.clone() for p in model.parameters()]

        gx_h, gp_h = grads(False)
        gx_t, gp_t = grads(True)
        np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-6, atol=1e-9 * float(gx_t.abs().max()), err_msg=str(c))
        for (name, _), a, b in zip(model.named_parameters(), gp_h, gp_t):
            np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-6, atol=1e-9 * max(1.0, float(b.abs().max())),
                                       err_msg=f"{c} {name}")


WIDE_BWD_CASES = [("pwquad", dict(d=16, n_cells=8, K=16, NN=[32, 32, 32])),          # C5 topology (matrix-core adjoint)
                  ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32, 32, 32])),          # C4 topology: P in {8,2,6,4,6,4,5,5}
                  ("pwlin", dict(d=8, P=4, n_cells=4, K=64, NN=[32, 32, 32], roll=4)),  # C3 topology
                  ("pwquad", dict(d=6, n_cells=3, K=7, NN=[16, 12])),
                  ("pwquad", dict(d=5, n_cells=2, K=3, NN=[20])),
                  ("pwlin", dict(d=7, P=3, n_cells=3, K=21, NN=[64, 48], roll=2)),
                  ("pwquad", dict(d=4, n_cells=2, K=9, NN=[40, 64, 33])),
                  ("pwquad", dict(d=8, n_cells=2, K=32, NN=[32, 32, 32])),            # golden cell shape (8, 4, 32, [32]*3)
                  ("pwquad", dict(d=5, n_cells=2, K=23, NN=[16])),
                  # heads whose four-column LDS ring does not fit: two / one column(s) per batch (CellK::mf_ncb in the adjoint head)
                  ("pwlin", dict(d=6, P=3, n_cells=2, K=128, NN=[16, 16], roll=2)),
                  # PWQuad with more than 32 bins behind a wide conditioner: the column's LDS ring doubles as the in-place dZ array
                  ("pwquad", dict(d=5, n_cells=2, K=40, NN=[16])),
                  ("pwquad", dict(d=4, n_cells=2, K=128, NN=[32, 32])),
                  ("pwquad", dict(d=3, n_cells=2, K=64, NN=[40, 64]))]


@pytest.mark.parametrize("kind,c", WIDE_BWD_CASES, ids=[f"{k}-d{c['d']}-K{c['K']}-NN{'x'.join(map(str, c['NN']))}" for k, c in WIDE_BWD_CASES])
def test_hip_backward_wide_models(kind, c, monkeypatch):
    """Matrix-core adjoint (flow_bwd_wm.hip: conditioner widths 9..64) against autograd through the stock-PyTorch
    re-evaluation of the same cells; N is not a multiple of the 16-sample tile."""
    from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
    torch.manual_seed(c["d"] * 7 + c["K"])
    if kind == "pwquad":
        NF = PWQuadManager(n_flow=c["d"])
        NF.create_model(c["n_cells"], c["K"], list(c["NN"]))
    else:
        NF = PWLinManager(n_flow=c["d"])
        NF.create_model(c["P"], c["n_cells"], c["K"], list(c["NN"]), c["roll"])
    model = NF._model.train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    assert model.engine().backward_supported(), c
    N, d = 1931, c["d"]
    x0 = torch.rand(N, d + 1, dtype=torch.double, device=dev()) * (1 - 1e-9)
    x0[:, d] += 0.5
    R = torch.randn(N, d + 1, dtype=torch.double, device=dev())

    def grads(force_torch):
        for p in model.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        out = model_torch(model, x) if force_torch else model(x)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]

    o_h, gx_h, gp_h = grads(False)
    o_t, gx_t, gp_t = grads(True)
    np.testing.assert_allclose(t2n(o_h), t2n(o_t), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-6, atol=1e-8 * float(gx_t.abs().max()), err_msg=str(c))
    for (name, _), a, b in zip(model.named_parameters(), gp_h, gp_t):
        np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-6, atol=1e-8 * max(1.0, float(b.abs().max())), err_msg=f"{c} {name}")


@pytest.mark.parametrize("kind,K,NN,n_cells", [("pwquad", 4, [3, 3, 3], 2), ("pwquad", 3, [4], 3), ("pwquad", 2, [2, 3], 2),
                                               ("pwlin", 7, [3, 3], 3), ("pwlin", 12, [4, 4, 4], 2)])
def test_hip_backward_matches_torch_adjoint(kind, K, NN, n_cells, monkeypatch):
    """The hand-written adjoint kernels (nis_flow_backward) against autograd through the stock-PyTorch re-evaluation,
    for a random upstream gradient on EVERY output column and a differentiable input."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    from nf_amd.normalizing_flows.layers.layers import RollLayer
    from nf_amd.normalizing_flows.sequential import FlowSequential
    torch.manual_seed(K * 10 + n_cells)
    cls = PWQuad if kind == "pwquad" else PWLin
    model = FlowSequential()
    for i in range(n_cells):
        model.add_module(str(i), cls(2, 1, K, list(NN)))
        model.add_module("roll" + str(i), RollLayer(1))
    model = model.double().to(dev()).train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.2 * torch.randn_like(p))
    assert model.engine().backward_supported()
    N = 3000
    x0 = torch.rand(N, 3, dtype=torch.double, device=dev())
    x0[:, 2] += 0.5
    if kind == "pwquad":
        x0[5, 1] = 1.0 - 1e-9                      # clamped sample: zero gradient through x
    else:
        x0[:, :2] *= (1 - 1e-12)
    R = torch.randn(N, 3, dtype=torch.double, device=dev())

    def grads(force_torch):
        for p in model.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        out = model_torch(model, x) if force_torch else model(x)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]

    o_h, gx_h, gp_h = grads(False)
    o_t, gx_t, gp_t = grads(True)
    np.testing.assert_allclose(t2n(o_h), t2n(o_t), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-7, atol=1e-9 * float(gx_t.abs().max()))
    for (name, _), a, b in zip(model.named_parameters(), gp_h, gp_t):
        np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-7, atol=1e-9 * max(1.0, float(b.abs().max())), err_msg=name)


@pytest.mark.parametrize("kind,d,P,K,NN,n_cells", [("pwquad", 2, 1, 4, [3, 3, 3], 2), ("pwlin", 3, 1, 7, [5, 4], 2), ("pwquad", 6, 3, 8, [16, 16], 2),
                                                   ("pwlin", 8, 4, 16, [32, 32, 32], 2), ("affine", 5, 2, 0, [8, 8], 3)])
def test_hip_backward_through_eval_mode_batchnorm(kind, d, P, K, NN, n_cells):
    """The reference's autograd differentiates a model in ANY mode; here the adjoint of a forward that ran eval-mode BatchNorm
    (running statistics: nis_flow_forward_hist / nis_flow_backward_eval -- no batch-mean terms) against autograd through the
    stock-PyTorch re-evaluation in eval mode: narrow (fp64 VALU) and wide (matrix-core) adjoint kernels, every cell kind."""
    from nf_amd.normalizing_flows.layers.coupling_cells import AffineCoupling, PWLin, PWQuad
    from nf_amd.normalizing_flows.layers.layers import RollLayer
    from nf_amd.normalizing_flows.sequential import FlowSequential
    torch.manual_seed(3 * d + n_cells)
    model = FlowSequential()
    for i in range(n_cells):
        cell = AffineCoupling(d, P, list(NN)) if kind == "affine" else (PWQuad if kind == "pwquad" else PWLin)(d, P, K, list(NN))
        model.add_module(str(i), cell)
        model.add_module("roll" + str(i), RollLayer(1))
    model = model.double().to(dev())
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.2 * torch.randn_like(p))
        model.train()
        for _ in range(3):                          # give the running statistics non-trivial values
            model(torch.cat((torch.rand(500, d, dtype=torch.double, device=dev()), torch.ones(500, 1, dtype=torch.double, device=dev())), 1))
    model.eval()
    assert model.engine().backward_supported()
    N = 2003
    x0 = torch.rand(N, d + 1, dtype=torch.double, device=dev())
    x0[:, :d] *= (1 - 1e-9)
    x0[:, d] += 0.5
    R = torch.randn(N, d + 1, dtype=torch.double, device=dev())
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k}

    def grads(force_torch):
        for p in model.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        out = model_torch(model, x) if force_torch else model(x)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]

    o_h, gx_h, gp_h = grads(False)
    for k, v in stats.items():                      # an eval-mode forward + backward leaves the buffers alone
        assert torch.equal(model.state_dict()[k], v), k
    o_t, gx_t, gp_t = grads(True)
    np.testing.assert_allclose(t2n(o_h), t2n(o_t), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-7, atol=1e-9 * float(gx_t.abs().max()))
    for (name, _), a, b in zip(model.named_parameters(), gp_h, gp_t):
        np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-7, atol=1e-9 * max(1.0, float(b.abs().max())), err_msg=name)


# ------------------------------------------------------------------------------ whole models
@pytest.mark.parametrize("case", golden("models.npz").cases())
def test_model_vs_reference_golden(case, g_models):
    m = g_models.meta[case]
    NF = build_manager(m, g_models.sub(case, "sd"))
    w = torch.as_tensor(g_models.get(case, "w")).to(dev())
    NF._model.train()
    with torch.no_grad():
        XJ = NF._model(NF.format_input(w, dev()))
    np.testing.assert_allclose(t2n(XJ), g_models.get(case, "XJ_train"), rtol=RTOL, atol=ATOL)
    sd = NF._model.state_dict()
    for k, v in g_models.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd[k]), v, rtol=1e-10, atol=1e-14, err_msg=k)
    NF._model.eval()
    with torch.no_grad():
        XJe = NF._model(NF.format_input(w, dev()))
    np.testing.assert_allclose(t2n(XJe), g_models.get(case, "XJ_eval"), rtol=RTOL, atol=ATOL)
    # the no-concat fast path returns the same numbers
    res = NF._model.flow_forward(w, want_xj=True)
    np.testing.assert_array_equal(t2n(res["xj"]), t2n(XJe))


# ------------------------------------------------------------------------------ integrate
class Replay:
    def __init__(self, draws):
        self.draws, self.i = draws, 0

    def __call__(self, n, d, dtype, stage):
        t = torch.as_tensor(self.draws[self.i])
        self.i += 1
        assert tuple(t.shape) == (n, d), (stage, tuple(t.shape), n, d)
        return t.to(dtype)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", golden("integrate.npz").cases())
def test_integrate_vs_reference_golden(case, fused, g_integrate):
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import normal
    m = g_integrate.meta[case]
    NF = build_manager(m, g_integrate.sub(case, "sd"))
    draws = g_integrate.get(case, "w")
    if case.startswith("camel"):
        f = integrands.Camel() if fused else (lambda x: integrands.Camel()(x))
    else:
        f = integrands.Normal(0.5, 0.25, 8) if fused else (lambda x: normal(x, 0.5, 0.25, 8))
    NF.sample_hook = Replay(draws)
    mom = t2n(NF.moments_batches(f, m["nitn"], m["neval"], dev()))
    np.testing.assert_allclose(mom[:, 1], g_integrate.get(case, "mean"), rtol=1e-9)
    np.testing.assert_allclose(mom[:, 2] / (mom[:, 0] - 1), g_integrate.get(case, "var"), rtol=1e-8)
    # full call on a fresh copy of the state (running stats advance with every train-mode forward)
    NF = build_manager(m, g_integrate.sub(case, "sd"))
    NF.sample_hook = Replay(draws)
    sig, sig_err = NF.integrate(f, m["nitn"], m["neval"], 0)
    assert sig.dtype == torch.float32 and sig.device.type == "cpu" and sig.dim() == 0
    np.testing.assert_allclose(float(sig), float(g_integrate.get(case, "sig")), rtol=2e-6)
    np.testing.assert_allclose(float(sig_err), float(g_integrate.get(case, "sig_err")), rtol=2e-6)
    sd = NF.best_model.state_dict()
    for k, v in g_integrate.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd[k]), v, rtol=1e-9, atol=1e-14, err_msg=k)


# ------------------------------------------------------------------------------ training
def camel_torch(x):
    return torch.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / (0.2 ** 2)) + \
        torch.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / (0.2 ** 2))


class RecordingAdamax(torch.optim.Adamax):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.seen = []

    def step(self, closure=None):
        self.seen.append([p.grad.detach().clone() for g in self.param_groups for p in g["params"]])
        return super().step(closure)


@pytest.mark.parametrize("case", golden("training.npz").cases())
def test_training_vs_reference_golden(case, g_training, tmp_path):
    m = g_training.meta[case]
    NF = build_manager(dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]), g_training.sub(case, "init/sd"))
    opt = RecordingAdamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
    draws = [g_training.get(case, f"draw{i:03d}") for i in range(m["n_draws"])]
    NF.sample_hook = Replay(draws)
    ret = NF._train_variance_forward_seq(camel_torch, opt, True, str(tmp_path), m["batch"], m["epochs"], 0, False, True,
                                         mini_batch_size=m["mbs"], integrate=True, preburn_time=m["preburn"],
                                         loss_mode=m["loss_mode"])
    hist = np.array([float(h) for h in NF.history])
    np.testing.assert_allclose(hist, g_training.get(case, "history"), rtol=1e-8)
    pnames = [k for k, _ in NF._model.named_parameters()]
    assert pnames == m["pnames"]
    for ei, grads in enumerate(opt.seen):
        for pn, g in zip(pnames, grads):
            ref = g_training.get(case, f"grad{ei}/{pn}")
            np.testing.assert_allclose(t2n(g), ref, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=f"{ei}/{pn}")
    np.testing.assert_allclose(np.array(ret), g_training.get(case, "ret"), rtol=1e-5)       # float32 estimator slots
    for att in ("best_loss", "varJ", "DKL", "int_loss", "best_loss_rel"):
        np.testing.assert_allclose(float(getattr(NF, att)), float(g_training.get(case, att)), rtol=1e-7, err_msg=att)
    np.testing.assert_allclose(float(NF.best_var), float(g_training.get(case, "best_var")), rtol=1e-6)
    assert int(NF.best_epoch) == int(g_training.get(case, "best_epoch"))
    assert float(NF.best_func_count) == float(g_training.get(case, "best_func_count"))
    for k, v in g_training.sub(case, "final/sd").items():
        np.testing.assert_allclose(t2n(NF._model.state_dict()[k]), v, rtol=1e-6, atol=1e-8, err_msg=k)   # atol: params whose gradient is pure rounding noise (BN bias in front of another BN)
    ck = torch.load(str(tmp_path / "torch"), weights_only=False)
    assert sorted(ck.keys()) == m["ckpt_keys"]


# ------------------------------------------------------------------------------ RAMBO
def make_ps(m):
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    return FlatInvertiblePhasespace(m["initial"], m["final"], pdf=None, pdf_active=m["pdf_active"], tau=m["tau"])


@pytest.mark.parametrize("levels", [0, 120])        # 0: Newton solver (default), 120: the reference's lattice bisection
@pytest.mark.parametrize("case", golden("rambo.npz").cases())
def test_rambo_vs_reference_golden(case, levels, g_rambo):
    m = g_rambo.meta[case]
    gen = make_ps(m)
    gen.bisect_levels = levels
    r = torch.as_tensor(g_rambo.get(case, "r")).to(dev())
    mom, wgt = gen.generateKinematics_batch(m["E_cm"], r, pT_mincut=m["cuts"][0], delR_mincut=m["cuts"][1],
                                            rap_maxcut=m["cuts"][2], pdgs=[2, -1])
    ref_m, ref_w = g_rambo.get(case, "momenta"), g_rambo.get(case, "weight")
    np.testing.assert_array_equal(t2n(wgt) == 0, ref_w == 0)                     # identical cut decisions
    np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-9, atol=1e-10 * np.abs(ref_m).max())
    np.testing.assert_allclose(t2n(wgt), ref_w, rtol=1e-9)


@pytest.mark.parametrize("n,massive,pdf", [(4, True, False), (6, False, True), (3, True, True), (8, False, False)])
def test_rambo_vs_oracle_seeded(n, massive, pdf):
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    rng = np.random.default_rng(n)
    final = [float(v) for v in (rng.uniform(5, 80, n) if massive else np.zeros(n))]
    init = [0., 0.]
    E = 13000. if pdf else 2000.
    N = 20011
    r = rng.random((N, 3 * n - 4 + (2 if pdf else 0)))
    cuts = (10., 0.4, 4.0) if pdf else (-1, -1, -1)
    want = {}
    ref_m, ref_w = orambo.generate_kinematics(init, final, E, r, *cuts, pdf_active=pdf, tau=True, want=want)
    gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=pdf, tau=True)
    mom, wgt = gen.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts)
    # cut decisions may legitimately differ only for points within rounding of a cut threshold
    differ = (t2n(wgt) == 0) != (ref_w == 0)
    assert differ.sum() <= 2
    ok = ~differ
    np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-8, atol=1e-9 * E)
    np.testing.assert_allclose(t2n(wgt)[ok], ref_w[ok], rtol=1e-8)


@pytest.mark.parametrize("n,massive,pdf", [(6, False, True), (4, True, False)])
def test_rambo_exact_zero_draws(n, massive, pdf):
    """A float32 draw is exactly 0.0 once per 1.7e7 values (integrate() draws float32, manager.py:390).  With v = 0 in an
    intermediate-mass column the reference's lattice bisection ends on 2^-levels > 0 and its weight stays finite; a root
    finder that returns u = 0 makes every K ratio 0/0 (the C5 leg of bench.py printed NaN).  Weights against the oracle."""
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    rng = np.random.default_rng(11 * n)
    final = [float(v) for v in (rng.uniform(5, 80, n) if massive else np.zeros(n))]
    E = 13000. if pdf else 2000.
    N = 4099
    r = rng.random((N, 3 * n - 4 + (2 if pdf else 0)))
    for c in range(n - 2):
        r[17 + 31 * c, c] = 0.0                                   # one exact zero in every mass column
    cuts = (10., 0.4, 4.0) if pdf else (-1, -1, -1)
    ref_m, ref_w = orambo.generate_kinematics([0., 0.], final, E, r, *cuts, pdf_active=pdf, tau=True)
    gen = FlatInvertiblePhasespace([0., 0.], final, pdf=None, pdf_active=pdf, tau=True)
    mom, wgt = gen.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts)
    w = t2n(wgt)
    assert np.isfinite(ref_w).all()
    assert np.isfinite(w).all()                                   # the point: no NaN leaks into the batch's moment records
    # the zero rows themselves are degenerate in the reference too (it carries NaN momenta through its cut comparisons, every
    # one of which is then False): only finiteness is asked of them; every other row matches the oracle as usual
    zero_rows = np.array([17 + 31 * c for c in range(n - 2)])
    keep = np.ones(N, dtype=bool)
    keep[zero_rows] = False
    same = ((w == 0) == (ref_w == 0)) & keep
    assert (keep & ~same).sum() <= 2
    np.testing.assert_allclose(w[same], ref_w[same], rtol=1e-8)
    # (massive case: the reference's own weight of such a row is an artefact of how many lattice levels the WHOLE batch needed
    #  -- rho(M_c, M_c+1, m_c) vanishes as u -> 0 and the 2^-levels floor decides what is left; nothing to match there)


def test_rambo_full_size_invariants():
    """BASELINE metric-ii shape: 2->4 massive, E_cm=1000, N=1e6: momentum conservation, on-shell legs,
    mean weight 0.02482, determinism."""
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
    g = torch.Generator(device="cuda").manual_seed(5)
    r = torch.rand(1_000_000, 8, dtype=torch.double, device=dev(), generator=g)
    mom, wgt = gen.generateKinematics_batch(1000., r, pT_mincut=0, delR_mincut=0, rap_maxcut=-1, pdgs=[2, -1])
    tot = mom[:, 2:].sum(1) - mom[:, :2].sum(1)
    assert float(tot.abs().max()) < 1e-9 * 1000.
    msq = mom[:, :, 0] ** 2 - (mom[:, :, 1:] ** 2).sum(-1)
    assert float((msq - 1e4).abs().max()) < 1e-6 * 1e4
    shat = 1000. ** 2
    mean_w = float(wgt.mean())          # the returned weight already carries 1/(2 shat)
    assert abs(mean_w - 0.02482) < 3e-4 * 3
    mom2, wgt2 = gen.generateKinematics_batch(1000., r, pT_mincut=0, delR_mincut=0, rap_maxcut=-1, pdgs=[2, -1])
    assert torch.equal(wgt, wgt2) and torch.equal(mom, mom2)


def test_rambo_nan_raises():
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace, PhaseSpaceGeneratorError
    gen = FlatInvertiblePhasespace([0.] * 2, [0.] * 3)
    r = torch.rand(100, 5, dtype=torch.double, device=dev())
    r[7, 2] = float("nan")
    with pytest.raises(PhaseSpaceGeneratorError):
        gen.generateKinematics_batch(100., r)


# ------------------------------------------------------------------------------ moments
@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 1000, 131072 + 17, 3_000_001])
def test_moments_vs_numpy(N):
    from nf_amd.flow import moments
    rng = np.random.default_rng(N)
    a = rng.normal(3.0, 0.01, N)                   # |mean| >> sigma: a naive sum of squares would lose digits
    out = t2n(moments(torch.as_tensor(a).to(dev())))
    assert out[0] == N
    np.testing.assert_allclose(out[1], a.mean(), rtol=1e-13)
    if N > 1:
        np.testing.assert_allclose(out[2] / (N - 1), a.var(ddof=1), rtol=1e-9)
    assert out[3] == a.max()
    b = rng.random(N)
    out = t2n(moments(torch.as_tensor(a).to(dev()), torch.as_tensor(b).to(dev())))
    np.testing.assert_allclose(out[1], (a * b).mean(), rtol=1e-12)


# ------------------------------------------------------------------------------ full-size properties
def test_c2_full_batch_properties():
    """C2 (2-D camel, PWQuad 2 cells, K=4, NN=[3]*3) at batch 1e6: every X in [0,1), mean J -> 1
    (a flow preserves volume), bitwise determinism, and shard/merge == whole batch in eval mode."""
    from nf_amd import integrands
    from nf_amd.flow import moments_of
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(0)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.rand(1_000_000, 2, dtype=torch.double, device=dev(), generator=g)
    NF._model.train()
    a = NF._model.flow_forward(w, want_xj=True)["xj"]
    assert float(a[:, :2].min()) >= 0.0 and float(a[:, :2].max()) < 1.0
    assert abs(float(a[:, 2].mean()) - 1.0) < 5e-3
    NF._model.eval()
    e1 = NF._model.flow_forward(w, want_xj=True)
    e2 = NF._model.flow_forward(w, want_xj=True)
    assert torch.equal(e1["xj"], e2["xj"])
    eng = NF._model.engine()
    whole, _ = moments_of(e1["planes"], 2, eng.out_col, integrands.Camel.nis_kind, integrands.Camel().nis_params)
    parts = []
    for s in range(4):
        ws = w[s * 250_000:(s + 1) * 250_000]
        res = NF._model.flow_forward(ws)
        parts.append(t2n(moments_of(res["planes"], 2, eng.out_col, integrands.Camel.nis_kind,
                                    integrands.Camel().nis_params)[0]))
    from nf_amd.parallel import merge_moments
    merged = merge_moments(np.stack(parts))
    np.testing.assert_allclose(merged[:3], t2n(whole)[:3], rtol=1e-11)
    assert merged[3] == t2n(whole)[3]
    # camel integral over the unit square (truncated Gaussians) = 0.2323 within MC error
    mean, var = merged[1], merged[2] / (merged[0] - 1)
    assert abs(mean - 0.2323235706) < 5 * np.sqrt(var / 1e6)       # 2 (s sqrt(pi)/2 (erf(1.25) + erf(3.75)))^2, s = 0.2


# ------------------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("N", [2, 3, 63, 65, 257, 1025])
def test_ragged_batch_sizes_vs_oracle(N):
    """Batch sizes around wave / workgroup / tile boundaries (N = 2 is the smallest train-mode batch torch accepts)."""
    torch.manual_seed(N)
    NF = build_manager(dict(kind="pwquad", d=5, n_cells=2, K=6, NN=[8, 8]))
    params = {k: t2n(v) for k, v in NF._model.state_dict().items()}
    w = torch.rand(N, 5, dtype=torch.double)
    steps = oflow.plan_pwquad(5, 2, 6)
    for train in (True, False):
        NF._model.train(train)
        ref = oflow.model_forward(w.numpy(), params, steps, train)
        got = NF._model.flow_forward(w.to(dev()), want_xj=True)["xj"]
        np.testing.assert_allclose(t2n(got), ref, rtol=1e-8 if N < 4 else RTOL, atol=1e-11)
        if train:                                   # running statistics moved: refresh the oracle's copy
            params = {k: t2n(v) for k, v in NF._model.state_dict().items()}


@pytest.mark.parametrize("N", [2, 65, 513, 131_073, 524_357, 700_001])
def test_folded_chain_batch_sizes_vs_oracle(N):
    """The d = 2 chain (single pass-through column: derived BN_1, folded input layer, k_bn_stats_fold with whole-tile and
    ragged-tail loops, prefetching SPEC apply kernels) at batch sizes around its tile boundaries: 256 workgroups x 512
    threads (131 072), 4 whole tiles per wave (524 288 + 64), and a ragged tail."""
    torch.manual_seed(N % 1000)
    NF = build_manager(dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]))
    params = {k: t2n(v) for k, v in NF._model.state_dict().items()}
    w = torch.rand(N, 2, dtype=torch.double)
    steps = oflow.plan_pwquad(2, 2, 4)
    NF._model.train()
    ref = oflow.model_forward(w.numpy(), params, steps, True)
    got = NF._model.flow_forward(w.to(dev()), want_xj=True)["xj"]
    np.testing.assert_allclose(t2n(got), ref, rtol=1e-8 if N < 4 else RTOL, atol=1e-11)
    # the fused integrand path of the same batch (SPEC kernels, nothing stored): moments of f(X) J
    from nf_amd import integrands
    f = integrands.Camel()
    NF2 = build_manager(dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]))
    NF2._model.load_state_dict({k: torch.as_tensor(v) for k, v in params.items()})
    NF2._model.train()
    out = torch.zeros((1, 4), dtype=torch.double, device=dev())
    NF2._model.engine().integrate_steps(w.to(dev()).unsqueeze(0), True, f.nis_kind, f.nis_params, out)
    fx = f(torch.as_tensor(ref[:, :2])).numpy() * ref[:, 2]
    m = t2n(out)[0]
    assert m[0] == N
    np.testing.assert_allclose(m[1], fx.mean(), rtol=1e-9)
    np.testing.assert_allclose(m[2], ((fx - fx.mean()) ** 2).sum(), rtol=1e-7)
    np.testing.assert_allclose(m[3], fx.max(), rtol=1e-9)
    # running statistics after one train-mode batch agree with the oracle's BatchNorm bookkeeping
    sdA, sdB = NF._model.state_dict(), NF2._model.state_dict()
    for k in sdA:
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(t2n(sdA[k]), t2n(sdB[k]), rtol=1e-10, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("n_cells,G,N", [(2, 1, 70_001), (2, 3, 40_003), (3, 2, 20_001), (4, 1, 131_073), (4, 2, 5)])
def test_direct_row_input_matches_planes_and_oracle(n_cells, G, N, monkeypatch):
    """integrate_steps on float32 draws of a d = 2 chain reads the (N, 2) rows of w inside the statistics / apply kernels
    (no plane ingest): same records and running statistics as the plane path, and the oracle's moments."""
    from nf_amd import integrands
    f = integrands.Camel()
    torch.manual_seed(100 + n_cells)
    cfg = dict(kind="pwquad", d=2, n_cells=n_cells, K=4, NN=[3, 3, 3])
    A = build_manager(cfg)
    sd = {k: v.clone() for k, v in A._model.state_dict().items()}
    B = build_manager(cfg)
    B._model.load_state_dict(sd)
    params = {k: t2n(v) for k, v in sd.items()}
    w = torch.rand(G, N, 2, dtype=torch.float32, device=dev())
    A._model.train()
    B._model.train()
    outA = torch.zeros((G, 4), dtype=torch.double, device=dev())
    A._model.engine().integrate_steps(w, True, f.nis_kind, f.nis_params, outA)                 # direct rows
    monkeypatch.setenv("NIS_NO_DIRECT_ROWS", "1")
    outB = torch.zeros((G, 4), dtype=torch.double, device=dev())
    B._model.engine().integrate_steps(w, True, f.nis_kind, f.nis_params, outB)                 # plane ingest
    monkeypatch.delenv("NIS_NO_DIRECT_ROWS")
    np.testing.assert_allclose(t2n(outA), t2n(outB), rtol=1e-13)
    for k in sd:
        np.testing.assert_allclose(t2n(A._model.state_dict()[k]), t2n(B._model.state_dict()[k]), rtol=1e-13, atol=1e-16, err_msg=k)
    steps = oflow.plan_pwquad(2, n_cells, 4)
    for g in range(G):                                                                         # every batch starts from the same BN state
        ref = oflow.model_forward(w[g].double().cpu().numpy(), params, steps, True)
        fx = f(torch.as_tensor(ref[:, :2])).numpy() * ref[:, 2]
        m = t2n(outA)[g]
        assert m[0] == N
        np.testing.assert_allclose(m[1], fx.mean(), rtol=1e-8)
        np.testing.assert_allclose(m[2], ((fx - fx.mean()) ** 2).sum(), rtol=1e-6)
        np.testing.assert_allclose(m[3], fx.max(), rtol=1e-8)


def test_single_sample_train_mode_raises_like_torch():
    NF = build_manager(dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]))
    NF._model.train()
    with pytest.raises(ValueError):
        NF._model.flow_forward(torch.rand(1, 2, dtype=torch.double, device=dev()))
    NF._model.eval()
    out = NF._model.flow_forward(torch.rand(1, 2, dtype=torch.double, device=dev()), want_xj=True)["xj"]
    assert out.shape == (1, 3) and torch.isfinite(out).all()


def test_pwlin_input_one_raises_index_error():
    """x == 1.0 selects bin n_bins: the reference's gather raises."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin
    cell = PWLin(3, 1, 8, [4, 4]).double().to(dev())
    x = torch.rand(100, 4, dtype=torch.double, device=dev())
    x[17, 2] = 1.0
    with pytest.raises(IndexError):
        cell(x)


@pytest.mark.parametrize("G,n_cells", [(1, 2), (3, 2), (7, 2), (2, 18)])
def test_batched_iterations_equal_one_by_one(G, n_cells):
    """nis_integrate_steps with G batches == G separate calls (records and running statistics), train mode.
    18 cells: the running-statistics epilogue (k_group_epilogue, 16 cells per launch) is split over two launches."""
    from nf_amd import integrands
    f = integrands.Camel()
    torch.manual_seed(4)
    A = build_manager(dict(kind="pwquad", d=2, n_cells=n_cells, K=4, NN=[3, 3, 3]))
    sd = {k: v.clone() for k, v in A._model.state_dict().items()}
    B = build_manager(dict(kind="pwquad", d=2, n_cells=n_cells, K=4, NN=[3, 3, 3]))
    B._model.load_state_dict(sd)
    w = torch.rand(G, 50_001, 2, dtype=torch.float32, device=dev())
    outA = torch.zeros((G, 4), dtype=torch.double, device=dev())
    A._model.engine().integrate_steps(w, True, f.nis_kind, f.nis_params, outA)
    outB = torch.zeros((G, 4), dtype=torch.double, device=dev())
    for g in range(G):
        B._model.engine().integrate_steps(w[g], True, f.nis_kind, f.nis_params, outB[g:g + 1])
    np.testing.assert_allclose(t2n(outA), t2n(outB), rtol=1e-12)
    for k in sd:
        np.testing.assert_allclose(t2n(A._model.state_dict()[k]), t2n(B._model.state_dict()[k]), rtol=1e-12, atol=1e-15, err_msg=k)


@pytest.mark.parametrize("cfg", ["massive8", "tau10"])
def test_flow_into_rambo_integrand_vs_oracle(cfg):
    """C4-style pipeline: flow output -> RAMBO weight * J -> moments, planar (fused) path == materialised path == oracle."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    if cfg == "massive8":
        d, init, final, pdf, E, cuts = 8, [100., 100.], [100.] * 4, False, 1000., (0, 0, -1)
    else:
        d, init, final, pdf, E, cuts = 10, [0., 0.], [100.] * 4, True, 13000., (10., 0.4, 4.0)
    torch.manual_seed(d)
    NF = build_manager(dict(kind="pwquad", d=d, n_cells=2, K=8, NN=[16, 16]))
    gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=pdf, tau=True)
    f = integrands.RamboWeight(gen, E, *cuts)
    N = 6001
    w = torch.rand(1, N, d, dtype=torch.float32)
    params = {k: t2n(v) for k, v in NF._model.state_dict().items()}
    steps = oflow.plan_pwquad(d, 2, 8)
    XJ = oflow.model_forward(w[0].numpy().astype(np.float64), params, steps, True)
    _, wref = orambo.generate_kinematics(init, final, E, XJ[:, :d], *cuts, pdf_active=pdf, tau=True)
    n, m, v, mx = oflow.moments(wref * XJ[:, d])
    NF.sample_hook = Replay([w[0].numpy()])
    mom_planar = t2n(NF.moments_batches(f, 1, N, dev()))[0]
    NF2 = build_manager(dict(kind="pwquad", d=d, n_cells=2, K=8, NN=[16, 16]), params)
    NF2.sample_hook = Replay([w[0].numpy()])
    mom_plain = t2n(NF2.moments_batches(lambda x: f(x), 1, N, dev()))[0]
    for got in (mom_planar, mom_plain):
        assert got[0] == N
        np.testing.assert_allclose(got[1], m, rtol=1e-8)
        np.testing.assert_allclose(got[2] / (N - 1), v, rtol=1e-7)


def test_data_parallel_training_two_ranks_one_gpu():
    """nf_amd.parallel.DataParallel under torch.distributed.run (2 ranks, gloo, both on this GPU): parameters, loss
    history and the returned estimate agree bit for bit across ranks although every rank sees its own minibatches."""
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIS_DIST_BACKEND="gloo", DP_EPOCHS="25", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "dp_train_check.py")],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_integrate_ranks_on_one_gpu_run_the_rccl_code_path(ranks):
    """The sample-sharded integrate() job with 2 and 3 ranks on this GPU (gloo serving the collective): every rank captures its launch
    set into a hipGraph inside the multi-rank job (no eager fallback), `allreduce_moments` takes the CUDA branch an RCCL job takes
    (ONE gather + ONE nis_merge_moments launch) and agrees with a NumPy Chan merge, and integrate_sharded is identical on all ranks."""
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "rank2_integrate_check.py")],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "R2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ------------------------------------------------------------------------------ AffineCoupling / AffineManager (8(f) rank 3)
@pytest.mark.parametrize("case", [c for c in golden("affine.npz").cases() if "model" not in c])
def test_affine_cell_vs_reference_golden(case, g_affine):
    from nf_amd.normalizing_flows.layers.coupling_cells import AffineCoupling
    m = g_affine.meta[case]
    cell = AffineCoupling(m["d"], m["P"], list(m["NN"])).double()
    sd = g_affine.sub(case, "sd")
    assert list(cell.state_dict().keys()) == list(sd.keys())            # biased hidden Linears, same key layout
    cell.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    cell = cell.to(dev()).train(m["mode"] == "train")
    x = torch.as_tensor(g_affine.get(case, "x")).to(dev())
    y = cell(x)
    np.testing.assert_allclose(t2n(y), g_affine.get(case, "y"), rtol=RTOL, atol=ATOL)
    sd_now = cell.state_dict()
    for k, v in g_affine.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd_now[k]), v, rtol=1e-11, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("case", [c for c in golden("affine.npz").cases() if "model" in c])
def test_affine_model_vs_reference_golden(case, g_affine):
    from nf_amd.normalizing_flows.manager import AffineManager
    m = g_affine.meta[case]
    NF = AffineManager(n_flow=m["d"])
    NF.create_model(m["P"], m["n_cells"], list(m["NN"]), m["roll"])
    model = NF._model
    assert [n + ":" + type(c).__name__ for n, c in model.named_children()] == m["children"]
    model.load_state_dict({k: torch.as_tensor(v) for k, v in g_affine.sub(case, "sd").items()})
    w = torch.as_tensor(g_affine.get(case, "w")).to(dev())
    model.train()
    XJ = model(NF.format_input(w, dev()))
    np.testing.assert_allclose(t2n(XJ), g_affine.get(case, "XJ_train"), rtol=RTOL, atol=ATOL)
    sd_now = model.state_dict()
    for k, v in g_affine.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd_now[k]), v, rtol=1e-11, atol=1e-14, err_msg=k)
    model.eval()
    XJe = model(NF.format_input(w, dev()))
    np.testing.assert_allclose(t2n(XJe), g_affine.get(case, "XJ_eval"), rtol=RTOL, atol=ATOL)


def test_affine_manager_trains_and_integrates():
    """AffineManager end to end: batched integrate (G iterations per launch) == one-by-one, and a few training epochs
    through the PyTorch-adjoint path decrease nothing to NaN; gradients match finite differences of the HIP forward."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import AffineManager
    torch.manual_seed(5)
    NF = AffineManager(n_flow=2)
    NF.create_model(1, 2, [3] * 3, 1)
    f = integrands.Camel()
    draws = [torch.rand(20000, 2, device=dev()).to(torch.float32) for _ in range(4)]
    it = iter(draws)
    NF.sample_hook = lambda n, d, dtype, stage: next(it).to(dtype)
    sd0 = {k: v.clone() for k, v in NF._model.state_dict().items()}
    NF.batch_iterations = 4
    a = NF.moments_batches(f, 4, 20000, dev()).cpu()
    NF._model.load_state_dict(sd0)
    it = iter(draws)
    NF.batch_iterations = 1
    b = NF.moments_batches(f, 4, 20000, dev()).cpu()
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-12)
    NF.sample_hook = None
    # gradient of one weight vs finite differences of the HIP forward
    cell = NF._model[0]
    x = torch.rand(4000, 3, dtype=torch.double, device=dev())
    x[:, 2] += 0.5
    out = cell(x)
    (out[:, -1].var() + out[:, 1].sum()).backward()
    p = cell.NN[1].weight
    eps = 1e-6
    with torch.no_grad():
        base = cell(x)
        p[0, 0] += eps
        pert = cell(x)
        p[0, 0] -= eps
    fd = ((pert[:, -1].var() + pert[:, 1].sum()) - (base[:, -1].var() + base[:, 1].sum())) / eps
    assert abs(float(fd) - float(p.grad[0, 0])) < 1e-4 * max(1.0, abs(float(fd)))
    optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
    res = NF._train_variance_forward_seq(f, optim, False, None, 4000, 12, 0, True, True, mini_batch_size=4000, integrate=True,
                                         preburn_time=3)
    assert all(np.isfinite(res)) and len(NF.history) >= 1 and all(bool(torch.isfinite(h)) for h in NF.history)
    # the trained model's integrate() body against the oracle on the same draw (the affine map does not cover the unit
    # cube, so the estimate itself is NOT the camel integral -- a property of the reference's cell)
    params = {k: t2n(v) for k, v in NF.best_model.state_dict().items()}      # integrate() samples the BEST model (:397)
    w = torch.rand(30000, 2, device=dev()).to(torch.float32)
    NF.sample_hook = lambda n, d, dtype, stage: w.to(dtype)
    NF.batch_iterations = 1
    mom = NF.moments_batches(f, 1, 30000, dev()).cpu().numpy()[0]
    XJ = oflow.model_forward(t2n(w).astype(np.float64), params, oflow.plan_affine(2, 1, 2, 1), True)
    fres = oflow.camel(XJ[:, :2]) * XJ[:, 2]
    np.testing.assert_allclose(mom[1], fres.mean(), rtol=1e-9)
    np.testing.assert_allclose(mom[2] / (mom[0] - 1), fres.var(ddof=1), rtol=1e-8)
    NF.sample_hook = None
    sig, err = NF.integrate(f, 5, 100000)
    assert np.isfinite(float(sig)) and float(err) > 0


def test_builtin_integrands_on_rows_match_the_torch_expression_and_the_oracle():
    """`nis_integrand_eval` (what the training loop's f(X) runs on) against the same formula in torch ops and NumPy, on
    the strided view XJ[:, :-1] the manager passes."""
    from nf_amd import integrands
    torch.manual_seed(3)
    for f, d in ((integrands.Camel(), 2), (integrands.Camel(0.7, 0.2, 0.3), 5), (integrands.Normal(0.5, 0.1, 8), 8),
                 (integrands.Normal(0.4, 0.3, 3), 3)):
        XJ = torch.rand(10007, d + 1, dtype=torch.double, device=dev())
        x = XJ[:, :-1]
        y = f(x)                                   # HIP kernel (float64 CUDA input, no grad)
        xr = x.clone().requires_grad_(True)
        y_t = f(xr)                                # differentiable input -> the torch expression
        assert y_t.requires_grad and not y.requires_grad
        np.testing.assert_allclose(t2n(y), t2n(y_t), rtol=1e-13, atol=1e-300)
        ref = oflow.camel(t2n(x)) if isinstance(f, integrands.Camel) and d == 2 else None
        if ref is not None:
            np.testing.assert_allclose(t2n(y), ref, rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("kind,K,NN", [("pwquad", 4, [3, 3, 3]), ("pwquad", 16, [8, 8]), ("pwlin", 12, [5]), ("pwquad", 9, [32, 32])])
def test_cell_is_a_monotone_bijection_of_the_unit_interval(kind, K, NN):
    """Known answers that need no reference: for a FIXED conditioner input every transformed column
    is a monotone map of [0,1) onto [0,1) whose derivative is the Jacobian factor, and the value at a bin edge is the
    cumulative area up to that bin."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    torch.manual_seed(K)
    cell = (PWQuad if kind == "pwquad" else PWLin)(2, 1, K, list(NN)).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.4 * torch.randn_like(p))
    cell = cell.to(dev()).eval()                                   # eval BN: the conditioner sees only xA
    N = 20001
    x = torch.empty(N, 3, dtype=torch.double, device=dev())
    x[:, 0] = 0.3712                                               # fixed pass-through value -> one fixed spline
    x[:, 1] = torch.linspace(0.0, 1.0 - 1e-6 if kind == "pwquad" else 1.0 - 1e-12, N, dtype=torch.double)
    x[:, 2] = 1.0
    y = cell(x)
    yb, jf = t2n(y[:, 1]), t2n(y[:, 2])
    xb = t2n(x[:, 1])
    assert yb[0] == 0.0 and abs(yb[-1] - 1.0) < 2e-5
    assert np.all(np.diff(yb) >= -1e-15) and np.all(jf > 0)
    # dy/dx == Jacobian factor (central differences away from the knots of the spline)
    num = (yb[2:] - yb[:-2]) / (xb[2:] - xb[:-2])
    rel = np.abs(num - jf[1:-1]) / jf[1:-1]
    assert np.quantile(rel, 0.98) < 5e-4
    # integral of the Jacobian over the column = 1 (trapezoid rule on the fine grid)
    # (PWLin's Jacobian is piecewise constant: the rule misplaces each of its K jumps by up to one grid step)
    assert abs(np.trapezoid(jf, xb) - yb[-1]) < (1e-6 if kind == "pwquad" else 1e-3)
    if kind == "pwlin":                                            # equal-width bins: the value at an edge k/K is the normalised prefix sum
        params = {"c." + k_: t2n(v) for k_, v in cell.state_dict().items()}
        Z = oflow.conditioner(np.array([[0.3712]]), params, "c", False).reshape(1, 1, K)
        Q = np.exp(Z[0, 0])
        edges = torch.tensor([[0.3712, k_ / K, 1.0] for k_ in range(K)], dtype=torch.double, device=dev())
        np.testing.assert_allclose(t2n(cell(edges)[:, 1]), np.concatenate([[0.0], np.cumsum(Q)[:-1]]) / Q.sum(), rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("d,P,NN,n_cells", [(2, 1, [3, 3, 3], 2), (5, 2, [8, 8], 3), (6, 4, [4], 2), (4, 1, [7, 5], 3),
                                            (8, 4, [32, 32, 32], 4), (7, 3, [16], 2), (9, 2, [20, 64], 3)])
def test_hip_backward_affine_matches_torch_adjoint(d, P, NN, n_cells, monkeypatch):
    """AffineCoupling models, narrow (fp64 VALU) and wide (matrix cores): hand-written adjoint (hidden-layer biases, [scale|shift][T] head) against autograd
    through the stock-PyTorch re-evaluation."""
    from nf_amd.normalizing_flows.manager import AffineManager
    torch.manual_seed(d * 11 + P)
    NF = AffineManager(n_flow=d)
    NF.create_model(P, n_cells, list(NN), 1)
    model = NF._model.train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.1 * torch.randn_like(p))
    assert model.engine().backward_supported()
    N = 2777
    x0 = torch.rand(N, d + 1, dtype=torch.double, device=dev())
    x0[:, d] += 0.5
    R = torch.randn(N, d + 1, dtype=torch.double, device=dev())

    def grads(force_torch):
        for p in model.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        out = model_torch(model, x) if force_torch else model(x)
        (out * R).sum().backward()
        return out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]

    o_h, gx_h, gp_h = grads(False)
    o_t, gx_t, gp_t = grads(True)
    np.testing.assert_allclose(t2n(o_h), t2n(o_t), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-7, atol=1e-9 * float(gx_t.abs().max()))
    for (name, _), a, b in zip(model.named_parameters(), gp_h, gp_t):
        np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-7, atol=1e-9 * max(1.0, float(b.abs().max())), err_msg=name)


def test_experiment_driver_reports_like_the_reference(tmp_path, monkeypatch):
    """`nisrep.utils.experiment_mg.pro(para)` (experiment_mg.py:12-124): same para keys, queue tuple and log.txt lines."""
    import queue as queue_mod
    from nf_amd import integrands
    from nisrep.utils import experiment_mg
    monkeypatch.setattr(experiment_mg, "epoch_length", 12, raising=False)
    import nf_amd.utils.experiment_mg as impl
    monkeypatch.setattr(impl, "epoch_length", 12)
    q = queue_mod.Queue()
    torch.manual_seed(1)
    para = dict(n_flow=2, n_bins=4, NN_width=3, NN_length=3, dev=0, lr=2e-3, weight_decay=1e-4, var_n=20000, batch_size=4000, pt=3,
                q=q, id=7, f=integrands.Camel(), logdir=str(tmp_path), log=True)
    impl.pro(para)
    res = q.get_nowait()
    assert len(res) == 14 and res[8] == "NIS" and res[10] == 7 and res[1] is None
    assert np.isfinite(res[11]) and np.isfinite(res[12]) and res[13] > 0
    text = (tmp_path / "7" / "log.txt").read_text()
    for needle in ("Started", "Create Model", "Start training", "End training", "Final Variance:", "Post training integrate:",
                   "Unweighting efficiency:", "Initial loss", "Best loss relative", "Evaluations", "Epoch"):
        assert needle in text, needle
    eff = float(text.split("Unweighting efficiency: ")[1].split("\n")[0])
    assert 0.0 < eff <= 1.0
    assert (tmp_path / "7" / "torch").exists() and (tmp_path / "7" / "torch_int").exists()     # the manager's checkpoints


@pytest.mark.parametrize("K,NN", [(4, [3, 3, 3]), (8, [3, 3]), (16, [2]), (5, [3, 3, 3]), (4, [4, 4])])
def test_batched_integrate_small_ragged_batches(K, NN):
    """Tiny / ragged batch sizes through the batched integrate path (two-column ingest, folded + 3-wide + compile-time
    specialised kernels, record merge in the consumers' prologue) against the oracle: moments of every batch and the
    running statistics after G train-mode batches applied in order."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(K)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, K, list(NN))
    with torch.no_grad():
        for p in NF._model.parameters():
            p.add_(0.2 * torch.randn_like(p))
    f = integrands.Camel()
    steps = oflow.plan_pwquad(2, 2, K)
    for train in (True, False):
        NF.best_model.train(train)
        for N in (2, 3, 63, 65, 257, 513, 1025, 4099):
            for G in (1, 3, 16):
                sd0 = {k: v.clone() for k, v in NF.best_model.state_dict().items()}
                cur = {k: t2n(v) for k, v in sd0.items()}
                w = torch.rand(G, N, 2, device=dev()).to(torch.float32)
                out = torch.zeros(G, 4, dtype=torch.double, device=dev())
                NF.best_model.engine().integrate_steps(w, train, f.nis_kind, f.nis_params, out)
                got = t2n(out)
                for g in range(G):
                    upd = {}
                    XJ = oflow.model_forward(t2n(w[g]).astype(np.float64), cur, steps, train, upd)
                    cur.update(upd)
                    v = oflow.camel(XJ[:, :2]) * XJ[:, 2]
                    ref = np.array([N, v.mean(), ((v - v.mean()) ** 2).sum(), v.max()])
                    np.testing.assert_allclose(got[g], ref, rtol=2e-9, atol=1e-12, err_msg=f"train={train} N={N} G={G} g={g}")
                if train:
                    sd1 = NF.best_model.state_dict()
                    for k_, v_ in cur.items():
                        if "running" in k_:
                            np.testing.assert_allclose(t2n(sd1[k_]), v_, rtol=1e-9, atol=1e-13, err_msg=f"{k_} N={N} G={G}")
                NF.best_model.load_state_dict(sd0)


@pytest.mark.parametrize("N", [1, 2, 3, 127, 128, 129, 257, 1001])
def test_rambo_small_and_ragged_batches_every_kernel_mode(N):
    """The four compile-time modes of k_rambo (plain, PDF, PDF + cuts, lattice bisection) and column-plane input on batch
    sizes around the 128-sample tile (never 4: reference quirk Q15) against the oracle."""
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    rng = np.random.default_rng(100 + N)
    for n, final, pdf, cuts, levels in ((4, [100.] * 4, False, (-1, -1, -1), 0), (4, [30., 10., 0., 5.], True, (0, 0, -1), 0),
                                        (5, [0.] * 5, True, (15., 0.4, 3.0), 0), (3, [20.] * 3, False, (-1, -1, -1), 120)):
        init = [0., 0.] if pdf else [100., 100.]
        E = 13000. if pdf else 1000.
        r = rng.random((N, 3 * n - 4 + (2 if pdf else 0)))
        ref_m, ref_w = orambo.generate_kinematics(init, final, E, r, *cuts, pdf_active=pdf, tau=True)
        gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=pdf, tau=True)
        gen.bisect_levels = levels
        mom, wgt = gen.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts)
        assert mom.shape == (N, n + 2, 4) and wgt.shape == (N,)
        same = (t2n(wgt) == 0) == (ref_w == 0)
        assert same.all() or N > 100 and (~same).sum() <= 1
        np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-8, atol=1e-9 * E, err_msg=f"n={n} pdf={pdf}")
        np.testing.assert_allclose(t2n(wgt)[same], ref_w[same], rtol=1e-8, err_msg=f"n={n} pdf={pdf}")


@pytest.mark.parametrize("ranks", [2, 3])
def test_exact_batchnorm_sharding_matches_the_whole_batch(ranks):
    """SURVEY.md 8(e) option 1: `FlowEngine.forward_sync` on the shards of a batch (torch.distributed.run, gloo, all ranks on
    this GPU) reproduces the single-process train-mode forward of the whole batch -- values and running statistics --
    for narrow, multi-column, PWLin and matrix-core models; `integrate_sharded(sync_bn=True)` agrees on every rank."""
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "syncbn_check.py")],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "SYNCBN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_wide_configs_full_batch_properties(name):
    """BASELINE configs C3 / C4 / C5 at their per-GPU batch sizes: every X in [0,1), J > 0 and finite,
    bitwise determinism in eval mode, shard / merge == whole batch (eval), and -- C4 / C5 -- the flow -> RAMBO weight
    integrand evaluated from the column planes equals `generateKinematics_batch` on the materialised X."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    from nf_amd.flow import moments as hip_moments, moments_of
    from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
    from nf_amd.parallel import merge_moments
    torch.manual_seed(3)
    if name == "C3":
        d, N = 8, 4_000_000
        NF = PWLinManager(n_flow=8)
        NF.create_model(4, 4, 64, [32] * 3, 4)
    elif name == "C4":
        d, N = 10, 500_000
        NF = PWQuadManager(n_flow=10)
        NF.create_model(5, 16, [32] * 3)
    else:
        d, N = 16, 1_000_000
        NF = PWQuadManager(n_flow=16)
        NF.create_model(8, 16, [32] * 3)
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.rand(N, d, dtype=torch.double, device=dev(), generator=g) * (1 - 1e-12)
    model = NF._model.train()
    a = model.flow_forward(w, want_xj=True)["xj"]
    assert float(a[:, :d].min()) >= 0.0 and float(a[:, :d].max()) <= 1.0
    assert bool(torch.isfinite(a).all()) and float(a[:, d].min()) > 0.0
    del a
    model.eval()
    e1 = model.flow_forward(w, want_xj=True)
    e2 = model.flow_forward(w, want_xj=True)
    assert torch.equal(e1["xj"], e2["xj"])
    eng = model.engine()
    f = integrands.Normal(0.5, 0.25, d)
    whole = t2n(moments_of(e1["planes"], d, eng.out_col, f.nis_kind, f.nis_params)[0])
    parts = []
    q = N // 4
    for s in range(4):
        res = model.flow_forward(w[s * q:(s + 1) * q])
        parts.append(t2n(moments_of(res["planes"], d, eng.out_col, f.nis_kind, f.nis_params)[0]))
    merged = merge_moments(np.stack(parts))
    np.testing.assert_allclose(merged[:3], whole[:3], rtol=1e-10)
    assert merged[3] == whole[3]
    if name in ("C4", "C5"):
        if name == "C4":
            gen = FlatInvertiblePhasespace([0., 0.], [100.] * 4, pdf=None, pdf_active=True)
            fw = integrands.RamboWeight(gen, 13000., 0, 0, -1, [2, -1])
        else:
            gen = FlatInvertiblePhasespace([0., 0.], [0.] * 6, pdf=None, pdf_active=True)
            fw = integrands.RamboWeight(gen, 13000., 20, 0.4, 2.5, [21, 21])
        assert gen.nDimPhaseSpace() + 2 == d
        from_planes = fw.from_planes(e1["planes"], eng.out_col, N)
        from_rows = fw(e1["xj"][:, :d].contiguous())
        assert torch.equal(from_planes, from_rows)               # same kernel, column-plane vs row-major input
        m = t2n(hip_moments(from_planes, e1["planes"][d]))
        assert np.isfinite(m).all() and m[0] == N and m[1] > 0
        if name == "C5":
            passed = float((from_planes > 0).double().mean())
            assert 0.05 < passed < 0.98                           # the cuts act (70.7 % pass for uniform points)


@pytest.mark.parametrize("G,N", [(1, 10000), (10, 200000)])
def test_integrate_steps_can_be_captured_in_a_hip_graph(G, N):
    """the entry points only enqueue on the caller's stream (no allocation, no host synchronisation), so a
    call can be captured into a hipGraph and replayed: bit-identical moments and running statistics."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(0)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    eng = NF.best_model.train().engine()
    f = integrands.Camel()
    w = torch.rand(G, N, 2, device=dev())
    out = torch.zeros(G, 4, dtype=torch.double, device=dev())
    ref = torch.zeros(G, 4, dtype=torch.double, device=dev())
    sd0 = {k: v.clone() for k, v in NF.best_model.state_dict().items()}
    tok = eng.pack_token()
    eng.integrate_steps(w, True, f.nis_kind, f.nis_params, ref, token=tok)          # eager reference (and cache warm-up)
    sd1 = {k: v.clone() for k, v in NF.best_model.state_dict().items()}
    NF.best_model.load_state_dict(sd0)
    tok = eng.pack_token()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
    torch.cuda.current_stream().wait_stream(side)
    NF.best_model.load_state_dict(sd0)
    tok = eng.pack_token()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
    NF.best_model.load_state_dict(sd0)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    sd2 = NF.best_model.state_dict()
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)


# ------------------------------------------------------------------------------ driver contract
def test_bench_prints_one_json_line_with_the_contract_fields():
    """`python bench.py` (short run, child process): exactly one JSON line on stdout carrying the driver's fields, the live
    roofline object of the dominant kernel and a result that agrees with the camel integral."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "10", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], typ), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 20 and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu_f64") and rf["unit"] in ("GB/s", "TFLOP/s", "Glane-instr/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["launches"] == 2 and rf["avg_launch_us"] > 0                      # the eager roofline set: 2 cells
    assert d["roofline_hbm"]["bound"] == "hbm" and d["config"]["steps_per_launch_set"] == 20
    for name in ("C3", "C4", "C5"):
        c = d["configs"][name]
        assert c["value"] > 0 and np.isfinite(c["integral"]["value"]) and c["roofline"]["bound"] == "mfma_f64"
    assert np.isfinite(d["rambo"]["integral"]["value"]) and abs(d["rambo"]["integral"]["value"] - 0.02482) < 2e-4
    assert abs(d["value"] - 1e6 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-9
    assert abs(d["integral"]["value"] - 0.2323235706) < 6 * d["integral"]["sigma_mc"]
    assert d["rambo"]["roofline"]["launches"] == 20


# ------------------------------------------------------------------------------ round 2: wide path pinned to the reference, full-size value checks
def kronecker_w(N, d):
    """tests/golden/gen_golden.py::kronecker_w (same integer arithmetic: the input of the N = 2e5 vectors is not stored)."""
    k = np.arange(1, N * d + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = (k * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(11)
    return (v.astype(np.float64) * 2.0 ** -53).reshape(N, d)


@pytest.fixture(scope="module")
def g_wide():
    return golden("wide.npz")


@pytest.mark.parametrize("tag", ["C3", "C4", "C5"])
def test_wide_forward_large_batch_vs_reference(tag, g_wide):
    """Matrix-core (MFMA f64) forward with train-mode BatchNorm at N = 200 000 -- many workgroups per BN layer, records merged
    across them -- against the unmodified reference's output on the same batch: 4 096 rows, per-column moments over all
    rows, and the running statistics the forward leaves behind."""
    case = f"fwd_{tag}"
    m = g_wide.meta[case]
    NF = build_manager(m, g_wide.sub(case, "sd"))
    model = NF._model
    model.train()
    w = torch.as_tensor(kronecker_w(m["N"], m["d"])).to(dev())
    XJ = model.flow_forward(w, want_xj=True)["xj"]
    rows = g_wide.get(case, "rows")
    np.testing.assert_allclose(t2n(XJ[torch.as_tensor(rows).to(dev())]), g_wide.get(case, "XJ_rows"), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(t2n(XJ.mean(0)), g_wide.get(case, "col_mean"), rtol=1e-10)
    np.testing.assert_allclose(t2n(XJ.var(0)), g_wide.get(case, "col_var"), rtol=1e-9)
    np.testing.assert_allclose(t2n(XJ.min(0).values), g_wide.get(case, "col_min"), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(t2n(XJ.max(0).values), g_wide.get(case, "col_max"), rtol=RTOL, atol=ATOL)
    sd = model.state_dict()
    for k, v in g_wide.sub(case, "sd_after").items():
        np.testing.assert_allclose(t2n(sd[k]), v, rtol=1e-10, atol=1e-13, err_msg=k)


def test_merge_of_gathered_moment_records_one_launch():
    """`nis_merge_moments` (what a multi-GPU job runs behind its one all-gather) against the NumPy sequential Chan merge: uneven
    and empty shards (count 0, max -inf), 37 iterations, 8 ranks."""
    from nf_amd.parallel import merge_gathered, merge_moments
    rng = np.random.default_rng(5)
    world, steps = 8, 37
    recs = np.zeros((world, steps, 4))
    for r in range(world):
        for i in range(steps):
            n = 0 if (r + i) % 5 == 0 else int(rng.integers(1, 2000))
            if n == 0:
                recs[r, i] = [0.0, 0.0, 0.0, -np.inf]
            else:
                x = rng.normal(3.0, 1e-2, n)
                recs[r, i] = [n, x.mean(), ((x - x.mean()) ** 2).sum(), x.max()]
    got = t2n(merge_gathered(torch.as_tensor(recs).to(dev())))
    ref = merge_moments(np.moveaxis(recs, 0, -2))
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=1e-12)
    np.testing.assert_array_equal(got[:, 3], ref[:, 3])


@pytest.mark.parametrize("tag", ["C3", "C5"])
def test_bn1_derived_from_the_pass_through_covariance(tag, g_wide, monkeypatch):
    """Matrix-core cells with 2..8 pass-through columns skip the layer-1 statistics launch: BN_0's records carry the co-moments of
    the block (k_bn_stats0_cov) and BN_1's batch mean / variance follow from h1 = W0 bn0(x) (load_coefs, CellK::dcov).  Against
    the measured path (NIS_NO_DCOV=1) on the same batch: values, Jacobians and every running statistic, G = 1 and G = 3."""
    case = f"fwd_{tag}"
    m = g_wide.meta[case]
    N = 50_000 + 37

    def run(no_dcov, G):
        if no_dcov:
            monkeypatch.setenv("NIS_NO_DCOV", "1")
        else:
            monkeypatch.delenv("NIS_NO_DCOV", raising=False)
        NF = build_manager(m, g_wide.sub(case, "sd"))
        model = NF._model
        model.train()
        eng = model.engine()
        w = torch.as_tensor(kronecker_w(G * N, m["d"]).reshape(G, N, m["d"])).to(dev())
        planes = torch.empty((G, eng.d + 1, N), dtype=torch.double, device=dev())
        out4 = torch.zeros((G, 4), dtype=torch.double, device=dev())
        from nf_amd import _lib
        eng.integrate_steps(w, True, _lib.NIS_F_ONE, [], out4, planes=planes)
        torch.cuda.synchronize()
        return t2n(planes), t2n(out4), {k: t2n(v) for k, v in model.state_dict().items()}

    for G in (1, 3):
        pa, oa, sa = run(False, G)
        pb, ob, sb = run(True, G)
        np.testing.assert_allclose(pa, pb, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(oa, ob, rtol=1e-10)
        for k in sb:
            np.testing.assert_allclose(sa[k], sb[k], rtol=1e-10, atol=1e-14, err_msg=k)


@pytest.mark.parametrize("tag", ["C3", "C5"])
def test_wide_adjoint_vs_reference_autograd(tag, g_wide):
    """The matrix-core adjoint kernels (flow_bwd_wm.hip) against the REFERENCE's autograd: gradients of every parameter and
    of the input for a fixed random cotangent on XJ, C3 / C5 topologies, train-mode BN, N = 512."""
    case = f"grad_{tag}"
    m = g_wide.meta[case]
    NF = build_manager(m, g_wide.sub(case, "sd"))
    model = NF._model
    assert model.engine().backward_supported()
    model.train()
    w = torch.as_tensor(g_wide.get(case, "w")).to(dev()).requires_grad_(True)
    cot = torch.as_tensor(g_wide.get(case, "cot")).to(dev())
    XJ = model(NF.format_input(w, dev()))
    np.testing.assert_allclose(t2n(XJ), g_wide.get(case, "XJ"), rtol=RTOL, atol=ATOL)
    (XJ * cot).sum().backward()
    np.testing.assert_allclose(t2n(w.grad), g_wide.get(case, "gw"), rtol=1e-6, atol=1e-9 * np.abs(g_wide.get(case, "gw")).max())
    ref = g_wide.sub(case, "grad")
    gmax = max(np.abs(v).max() for v in ref.values())
    for pn, p in model.named_parameters():
        # atol: a bias in front of a train-mode BatchNorm has an analytically ZERO gradient -- both sides hold rounding noise there
        np.testing.assert_allclose(t2n(p.grad), ref[pn], rtol=1e-6, atol=1e-10 * gmax, err_msg=pn)


def _oracle_c2(w, params):
    """Oracle forward of the C2 model cell by cell (bins of every cell kept)."""
    steps = oflow.plan_pwquad(2, 2, 4)
    x = np.concatenate([np.asarray(w, dtype=np.float64), np.ones((w.shape[0], 1))], axis=1)
    cur, bins, upd_all = dict(params), [], {}
    for st in steps:
        if st[0] == "cell":
            want, upd = {}, {}
            x = oflow.cell_forward(x, cur, st[1], st[2], st[3], st[4], True, upd, want)
            cur.update(upd)
            upd_all.update(upd)
            bins.append(want["bins"][:, 0])
        else:
            x = np.concatenate([np.roll(x[:, :2], st[1], axis=1), x[:, 2:]], axis=1)
    return x, np.stack(bins), upd_all


def test_c2_full_batch_values_and_bins_vs_oracle():
    """BASELINE config C2 at its full batch (1e6 float32 draws), on the kernels bench.py times: the moment record of
    camel(X) * J against the oracle's VALUES (not properties), the bins the SPEC kernels select bit-exact (test
    instantiation that writes them), and the running statistics."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(21)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    model = NF.best_model
    model.train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.2 * torch.randn_like(p))
    params = {k: t2n(v) for k, v in model.state_dict().items()}
    N = 1_000_000
    w = torch.rand(N, 2, dtype=torch.float32, device=dev())
    w[:4, 0] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 0.5, 0.25])               # lattice edge values
    XJ_ref, bins_ref, upd = _oracle_c2(t2n(w), params)
    fres = oflow.camel(XJ_ref[:, :2]) * XJ_ref[:, 2]
    n, mean, var, mx = oflow.moments(fres)
    f = integrands.Camel()
    eng = model.engine()
    rec, bins = eng.integrate_step_bins(w, f.nis_kind, f.nis_params)
    rec = t2n(rec)
    np.testing.assert_array_equal(t2n(bins), bins_ref)
    assert rec[0] == n
    np.testing.assert_allclose(rec[1], mean, rtol=1e-11)
    np.testing.assert_allclose(rec[2] / (n - 1), var, rtol=1e-9)
    np.testing.assert_allclose(rec[3], mx, rtol=1e-9)
    sd = model.state_dict()
    for k, v in upd.items():
        np.testing.assert_allclose(t2n(sd[k]), v, rtol=1e-10, atol=1e-14, err_msg=k)
    # the production entry (no bin output) gives the same record bit for bit
    model.load_state_dict({k: torch.as_tensor(v) for k, v in params.items()})
    out = torch.zeros((1, 4), dtype=torch.double, device=dev())
    eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    np.testing.assert_array_equal(t2n(out[0]), rec)


@pytest.mark.parametrize("N", [20001, 20002])          # odd: float32 rows; even: two float32 column planes (PLN kernel instantiations)
def test_in_kernel_draws_match_documented_mapping_and_oracle(N):
    """nis_integrate_steps_rng: the draws are exactly Philox4x32-10 in the documented mapping (NumPy restatement with the
    Random123 known-answer vector), and the records equal the oracle's on those draws."""
    from nf_amd import integrands
    from nf_amd.flow import philox_rows
    from nf_amd.normalizing_flows.manager import PWQuadManager

    def philox(ctr, key):                                           # Salmon et al. 2011, 10 rounds, uint32 arrays
        c = [np.asarray(x, dtype=np.uint64) for x in ctr]
        k0, k1 = np.uint64(key[0]), np.uint64(key[1])
        M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
            k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
        return c

    kat = philox([0, 0, 0, 0], [0, 0])
    assert [int(x) for x in kat] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]      # Random123 kat_vectors, philox4x32 10
    G, seed, s0 = 3, 0x1234567890ABCDEF, 5
    rows = t2n(philox_rows(seed, s0, G, N, 2, dev()))
    for g in range(G):
        c = np.arange((N + 1) // 2, dtype=np.uint64)
        sid = np.uint64(s0 + g)
        r = philox([c & np.uint64(0xFFFFFFFF), c >> np.uint64(32), np.full_like(c, sid & np.uint64(0xFFFFFFFF)), np.full_like(c, sid >> np.uint64(32))],
                   [seed & 0xFFFFFFFF, seed >> 32])
        want = (np.stack(r, 1) >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
        np.testing.assert_array_equal(rows[g].reshape(-1), want.reshape(-1)[:2 * N])
    torch.manual_seed(4)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    model = NF.best_model
    model.train()
    params = {k: t2n(v) for k, v in model.state_dict().items()}
    f = integrands.Camel()
    out = torch.zeros((G, 4), dtype=torch.double, device=dev())
    wout = torch.zeros((G, N, 2), dtype=torch.float32, device=dev())
    model.engine().integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, out, seed, s0, w_out=wout)
    np.testing.assert_array_equal(t2n(wout), rows)
    sig, sig_err, means, variances, _ = oflow.integrate(oflow.camel, params, oflow.plan_pwquad(2, 2, 4), list(rows))
    rec = t2n(out)
    np.testing.assert_allclose(rec[:, 1], means, rtol=1e-10)
    np.testing.assert_allclose(rec[:, 2] / (rec[:, 0] - 1), variances, rtol=1e-9)


def test_graph_replay_of_a_launch_set_equals_eager_launches():
    """hipGraph replay of a launch set gives the records of the plain launch-per-phase path bit for bit, the same BatchNorm
    running statistics, and the same records as feeding the materialised draws."""
    import copy
    from nf_amd import integrands
    from nf_amd.flow import philox_rows
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(9)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    base = NF.best_model
    base.train()
    f = integrands.Camel()
    G, N = 7, 300_017

    def run(how):
        model = copy.deepcopy(base)
        eng = model.engine()
        out = torch.zeros((G, 4), dtype=torch.double, device=dev())
        if how == "graph":
            plan = eng.plan(G, N, dev(), f.nis_kind, f.nis_params)
            assert plan.captured, "the launch set must be captured, not fall back to eager launches"
            plan.reset(77, 3)
            plan.run(out, replay=True)
        elif how == "fed":
            eng.integrate_steps(philox_rows(77, 3, G, N, 2, dev()), True, f.nis_kind, f.nis_params, out)
        else:
            eng.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, out, 77, 3)
        torch.cuda.synchronize()
        return t2n(out), {k: t2n(v) for k, v in model.state_dict().items()}

    ref, sd_ref = run("rng")
    got, sd = run("graph")
    np.testing.assert_array_equal(got, ref)
    for k in sd_ref:
        np.testing.assert_allclose(sd[k], sd_ref[k], rtol=1e-9, atol=1e-15, err_msg=k)
    fed, _ = run("fed")
    np.testing.assert_allclose(fed, ref, rtol=1e-12)


@pytest.mark.parametrize("n_cells,N", [(3, 20002), (4, 50_000), (3, 20001)])
def test_in_kernel_draws_longer_d2_chains_match_the_fed_path(n_cells, N):
    """Chains of 3 and 4 cells with in-kernel draws: an INNER cell overwrites the state plane that also holds the float32 draws, so
    the draws must stay in the row layout there (a thread overwrites exactly the row it has read; the two-plane layout is for the
    2-cell chain only).  Records must equal those of feeding the materialised draws, even and odd N."""
    from nf_amd import integrands
    from nf_amd.flow import philox_rows
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(21)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(n_cells, 4, [3] * 3)
    model = NF.best_model
    model.train()
    eng = model.engine()
    f = integrands.Camel()
    G, seed, s0 = 3, 99, 11
    assert eng.rng_mode(G, N, True, f.nis_kind) == 2
    out = torch.zeros((G, 4), dtype=torch.double, device=dev())
    wout = torch.zeros((G, N, 2), dtype=torch.float32, device=dev())
    eng.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, out, seed, s0, w_out=wout)
    rows = philox_rows(seed, s0, G, N, 2, dev())
    assert torch.equal(wout, rows)
    fed = torch.zeros((G, 4), dtype=torch.double, device=dev())
    eng.integrate_steps(rows, True, f.nis_kind, f.nis_params, fed)
    np.testing.assert_allclose(t2n(out), t2n(fed), rtol=1e-12)
    params = {k: t2n(v) for k, v in model.state_dict().items()}
    _, _, means, variances, _ = oflow.integrate(oflow.camel, params, oflow.plan_pwquad(2, n_cells, 4), list(t2n(rows)[:1]))
    # (the oracle ran on the parameters AFTER the calls above: running statistics do not enter train-mode values)
    np.testing.assert_allclose(t2n(out)[0, 1], means[0], rtol=1e-10)


def _np_philox(ctr, key):
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return c


@pytest.mark.parametrize("kind,c", [("pwlin", dict(d=8, P=4, n_cells=4, K=64, NN=[32, 32, 32], roll=4)),
                                    ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32, 32, 32])),
                                    ("pwquad", dict(d=5, n_cells=5, K=6, NN=[7, 7])),
                                    ("pwquad", dict(d=3, n_cells=3, K=4, NN=[3, 3, 3]))])
@pytest.mark.parametrize("train", [True, False])
def test_in_kernel_draws_any_model_match_mapping_and_fed_path(kind, c, train):
    """nis_integrate_steps_rng for models outside the d = 2 direct-row class: the ingest launch GENERATES the float64 planes.  The
    draws are Philox4x32-10 in the documented d != 2 mapping (NumPy restatement), and the records equal those of feeding the same
    draws through nis_integrate_steps -- including the matrix-core first cell whose BN_0 records now carry the co-moments (BN_1
    derived, no covariance pass, no layer-1 statistics launch)."""
    from nf_amd import integrands
    from nf_amd.flow import philox_rows
    torch.manual_seed(31)
    m = dict(c, kind=kind)
    NF = build_manager(m)
    model = NF.best_model
    model.train(train)
    eng = model.engine()
    d = m["d"]
    f = integrands.Normal(0.5, 0.2, d)
    G, N, seed, s0 = 2, 6001, 0xABCDEF0123, 7
    assert eng.rng_mode(G, N, train, f.nis_kind) == 1
    rows = philox_rows(seed, s0, G, N, d, dev())
    Q = (d + 3) // 4
    for g in range(G):
        ctr = (np.arange(N, dtype=np.uint64)[:, None] * np.uint64(Q) + np.arange(Q, dtype=np.uint64)[None, :]).reshape(-1)
        sid = np.uint64(s0 + g)
        r = _np_philox([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.full_like(ctr, sid & np.uint64(0xFFFFFFFF)),
                        np.full_like(ctr, sid >> np.uint64(32))], [seed & 0xFFFFFFFF, seed >> 32])
        want = ((np.stack(r, 1) >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(N, 4 * Q)[:, :d]
        np.testing.assert_array_equal(t2n(rows[g]), want)
    out = torch.zeros((G, 4), dtype=torch.double, device=dev())
    wout = torch.zeros((G, N, d), dtype=torch.float32, device=dev())
    eng.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, out, seed, s0, w_out=wout, train=train)
    assert torch.equal(wout, rows)
    fed = torch.zeros((G, 4), dtype=torch.double, device=dev())
    eng.integrate_steps(rows, train, f.nis_kind, f.nis_params, fed)
    np.testing.assert_allclose(t2n(out), t2n(fed), rtol=1e-9)
    # kept planes: X, J of the rng path equal the fed path's
    pl_a = torch.zeros((G, d + 1, N), dtype=torch.double, device=dev())
    pl_b = torch.zeros_like(pl_a)
    eng.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, out, seed, s0, train=train, planes=pl_a, keep_planes=True)
    eng.integrate_steps(rows, train, f.nis_kind, f.nis_params, fed, planes=pl_b)
    np.testing.assert_allclose(t2n(pl_a), t2n(pl_b), rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("loss_mode", ["var", "est"])
@pytest.mark.parametrize("builtin", [True, False])
@pytest.mark.parametrize("shape", [dict(d=2, n_cells=2, K=4, NN=[3, 3, 3]), dict(d=9, n_cells=7, K=6, NN=[16, 16])])
def test_fused_training_minibatch_equals_the_piecewise_path(loss_mode, builtin, shape):
    """nis_loss_terms + nis_flow_backward_loss (forward, integrand, loss bookkeeping and adjoint as library calls) against the
    piecewise path through torch's autograd graph (nis_moments + elementwise torch ops + nis_flow_backward), which the golden
    training runs pin to the reference: same draws -> same loss history, estimator bookkeeping, gradients and parameters."""
    import copy
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    d = shape["d"]
    torch.manual_seed(77)
    NF0 = PWQuadManager(n_flow=d)
    NF0.create_model(shape["n_cells"], shape["K"], shape["NN"])
    g = torch.Generator().manual_seed(5)
    draws = [torch.rand(4096, d, dtype=torch.double, generator=g) for _ in range(64)]
    if builtin:
        f = integrands.Camel() if d == 2 else integrands.Normal(0.5, 0.3, d)
    else:
        def f(x):                                   # an arbitrary user callable
            return torch.exp(-((x - 0.4) ** 2).sum(-1) * 3.0) + 0.1

    def run(fused):
        NF = PWQuadManager(n_flow=d)
        NF._model = copy.deepcopy(NF0._model)
        NF.best_model = NF._model
        NF.fused_minibatch = fused
        it = iter(draws)
        NF.sample_hook = lambda n, dd, dtype, stage: next(it)[:n].to(dtype)
        optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        res = NF._train_variance_forward_seq(f, optim, False, None, 2048, 6, 0, True, True, mini_batch_size=1024, integrate=True,
                                             preburn_time=3, loss_mode=loss_mode)
        grads = [t2n(p.grad) for p in NF._model.parameters()]
        return res, t2n(torch.stack(NF.history)), grads, [t2n(p) for p in NF._model.parameters()]

    ra, ha, ga, pa = run(True)
    rb, hb, gb, pb = run(False)
    np.testing.assert_allclose(ha, hb, rtol=1e-9)
    np.testing.assert_allclose(ra, rb, rtol=1e-6)
    for x, y in zip(ga, gb):
        np.testing.assert_allclose(x, y, rtol=1e-7, atol=1e-12)
    for x, y in zip(pa, pb):               # atol: parameters whose gradient is pure rounding noise (BN_0's shift cancels in BN_1) take
        np.testing.assert_allclose(x, y, rtol=1e-7, atol=1e-6)      # lr-sized Adamax steps of arbitrary sign, as in the golden training tests


def test_a_captured_launch_set_survives_other_engine_calls():
    """The IntegratePlan owns its workspace and state planes: engine calls with a LARGER set size, another batch size or another
    BatchNorm mode (which re-allocate / re-pack the engine's own buffers with a different layout) must not disturb a later replay."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(5)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3)
    model = NF.best_model
    model.train()
    eng = model.engine()
    f = integrands.Camel()
    G, N = 4, 100_003
    plan = eng.plan(G, N, dev(), f.nis_kind, f.nis_params)
    assert plan.captured
    first = torch.zeros((G, 4), dtype=torch.double, device=dev())
    plan.reset(5, 0)
    plan.run(first, replay=True)
    # calls that used to move / re-layout what the graph pointed at
    big = torch.zeros((3 * G + 1, 4), dtype=torch.double, device=dev())
    eng.integrate_steps_rng(3 * G + 1, 2 * N + 7, dev(), f.nis_kind, f.nis_params, big, 1, 0)
    eng.integrate_steps(torch.rand(2, 777, 2, device=dev()), False, f.nis_kind, f.nis_params, big[:2])
    junk = [torch.full((1 << 22,), 7.0, device=dev()) for _ in range(8)]      # recycle whatever the allocator freed
    assert eng.plan(G, N, dev(), f.nis_kind, f.nis_params) is plan
    again = torch.zeros((G, 4), dtype=torch.double, device=dev())
    plan.reset(5, 0)
    plan.run(again, replay=True)
    np.testing.assert_array_equal(t2n(again), t2n(first))
    eager = torch.zeros((G, 4), dtype=torch.double, device=dev())
    eng.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, eager, 5, 0)
    np.testing.assert_array_equal(t2n(eager), t2n(first))
    del junk


@pytest.mark.parametrize("cfg", ["massive8", "tau10", "cuts16"])
def test_fused_rambo_tail_equals_weight_times_jacobian_moments(cfg):
    """planes -> RAMBO weight * J -> moment records in ONE launch per set (nis_rambo_moments) against the separate weight kernel +
    nis_moments per batch, on the kept planes of a real flow; and BasicManager.integrate() through that path against the
    per-iteration path on identical draws."""
    from nf_amd import integrands
    from nf_amd.flow import moments as hip_moments, philox_rows
    from nf_amd.normalizing_flows.manager import PWQuadManager
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    torch.manual_seed(17)
    if cfg == "massive8":
        d, gen, args = 8, FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False), (1000., -1, -1, -1, [0, 0])
    elif cfg == "tau10":
        d, gen, args = 10, FlatInvertiblePhasespace([0., 0.], [100.] * 4, pdf=None, pdf_active=True), (13000., 0, 0, -1, [2, -1])
    else:
        d, gen, args = 16, FlatInvertiblePhasespace([0., 0.], [0.] * 6, pdf=None, pdf_active=True), (13000., 20, 0.4, 2.5, [21, 21])
    NF = PWQuadManager(n_flow=d)
    NF.create_model(2 * int(np.ceil(np.log2(d))), 8, [16, 16])
    model = NF.best_model
    model.train()
    eng = model.engine()
    f = integrands.RamboWeight(gen, args[0], args[1], args[2], args[3], args[4])
    G, N = 3, 5003
    scratch = torch.zeros((G, 4), dtype=torch.double, device=dev())
    planes = torch.zeros((G, d + 1, N), dtype=torch.double, device=dev())
    eng.integrate_steps_rng(G, N, dev(), 2, [], scratch, 3, 0, planes=planes, keep_planes=True)
    out = torch.zeros((G, 4), dtype=torch.double, device=dev())
    status = torch.zeros(1, dtype=torch.int32, device=dev())
    f.moments_from_planes(planes, eng.out_col, d, N, out, status)
    assert not f.check_status(status)
    for g in range(G):
        want = hip_moments(f.from_planes(planes[g], eng.out_col, N), planes[g][d])
        np.testing.assert_allclose(t2n(out[g]), t2n(want), rtol=1e-12)
    # integrate(): in-kernel draws + fused tail vs the per-iteration path fed with the same draws
    nitn = 4
    torch.manual_seed(123)
    sig, err = NF.integrate(f, nitn, N)
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    rows = philox_rows(seed, 0, nitn, N, d, dev())
    it = iter(range(nitn))
    NF.sample_hook = lambda n, dd, dtype, stage: rows[next(it)].to(dtype)
    NF.batch_iterations = 1
    try:
        sig2, err2 = NF.integrate(f, nitn, N)
    finally:
        NF.sample_hook = None
    np.testing.assert_allclose(float(sig), float(sig2), rtol=2e-6)
    np.testing.assert_allclose(float(err), float(err2), rtol=2e-6)


# ------------------------------------------------------------------------------ device-side PDF interpolation (parity unpinned)
def _synthetic_grid(two=False):
    from test_oracle_golden import synthetic_pdf_grid
    from oracle import pdf as opdf
    from nf_amd.PhaseSpace.pdf_grid import PdfGrid
    subs, f = synthetic_pdf_grid(two)
    return subs, PdfGrid.from_lhagrid1(opdf.write_lhagrid1(subs))


@pytest.mark.parametrize("two", [False, True])
def test_pdf_grid_xfx_vs_oracle(two):
    """PdfGrid.xfxQ2 (k_pdf_xfx: 1-D log-cubic on the host-folded table) against the NumPy restatement of LHAPDF's
    log-bicubic rule: interior, knots, both grid edges, arguments outside the grid (frozen at the boundary)."""
    from oracle import pdf as opdf
    subs, grid = _synthetic_grid(two)
    rng = np.random.default_rng(5)
    x = np.concatenate([10 ** rng.uniform(-6, 0, 20000), subs[0].xs, [1e-7, 1e-9, 1.0, 1.5, subs[0].xs[0], subs[0].xs[-1]]])
    for q2 in (91.188 ** 2, 20.0, 99.9, 100.0, 5e7, 1.0, 1e9):
        for pid in (21, -2, 1):
            got = t2n(grid.xfxQ2(pid, torch.as_tensor(x).to(dev()), q2))
            want = opdf.xfx_q2(subs, pid, x, q2)
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-14, err_msg=f"{pid} {q2}")


@pytest.mark.parametrize("n,cuts,pdgs", [(4, (-1, -1, -1), [21, 21]), (6, (20., 0.4, 2.5), [2, -1]), (3, (5., 0.3, -1), [22, 21])])
def test_rambo_with_pdf_grid_vs_oracle(n, cuts, pdgs):
    """generateKinematics_batch with a PDF grid: x f(x, 91.188^2) / x per beam evaluated inside k_rambo (reference :120-137,
    :183-187) -- weights against the oracle (pdf=None weight x the NumPy LHAPDF-rule densities), row and plane input."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    subs, grid = _synthetic_grid(True)
    rng = np.random.default_rng(n)
    final = [0.0] * n if n != 4 else [float(v) for v in rng.uniform(5, 80, n)]
    E, N = 13000., 20011
    r = rng.random((N, 3 * n - 2))
    ref_m, ref_w = orambo.generate_kinematics([0., 0.], final, E, r, *cuts, pdf_active=True, tau=True, pdf=subs, pdgs=pdgs)
    gen = FlatInvertiblePhasespace([0., 0.], final, pdf=grid, pdf_active=True, tau=True)
    mom, wgt = gen.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts, pdgs=pdgs)
    w = t2n(wgt)
    differ = (w == 0) != (ref_w == 0)
    assert differ.sum() <= 2
    np.testing.assert_allclose(w[~differ], ref_w[~differ], rtol=1e-8)
    np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-8, atol=1e-9 * E)
    # the densities really entered: the same points without the grid differ by exactly that factor
    gen0 = FlatInvertiblePhasespace([0., 0.], final, pdf=None, pdf_active=True, tau=True)
    _, w0 = gen0.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts, pdgs=pdgs)
    ratio = w[(w != 0)] / t2n(w0)[(w != 0)]
    assert ratio.std() > 0.1 * abs(ratio.mean()) and np.isfinite(ratio).all()
    # plane input (the flow's layout): RamboWeight.from_planes
    planes = torch.as_tensor(r.T.copy()).to(dev())
    f = integrands.RamboWeight(gen, E, *cuts, pdgs=pdgs)
    wp = t2n(f.from_planes(planes, list(range(3 * n - 2)), N))
    np.testing.assert_array_equal(wp, w)


def test_engines_sharing_cells_see_each_others_running_statistics():
    """`model[0](x)` runs through the cell's own engine, `model(x)` through the Sequential's: a train-mode forward of one
    advances the BatchNorm running statistics, and an eval-mode forward of the OTHER must then use them (the epoch that
    invalidates a packed eval-mode image is kept on the cell, not on the engine)."""
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(5)
    NF = PWQuadManager(n_flow=3)
    NF.create_model(3, 5, [6, 6])
    model = NF._model
    x = torch.rand(4000, 4, dtype=torch.double, device=dev())
    x[:, 3] = 1.0
    with torch.no_grad():
        model.eval()
        y0 = model(x).clone()                                  # packs the Sequential engine's eval-mode image
        model.train()
        for _ in range(3):
            model[0](torch.rand(3000, 4, dtype=torch.double, device=dev()))      # cell engine: running stats of cell 0 move
        model.eval()
        y1 = model(x)
    params = {k: t2n(v) for k, v in model.state_dict().items()}
    ref = oflow.model_forward(t2n(x[:, :3]), params, oflow.plan_pwquad(3, 3, 5), False)
    np.testing.assert_allclose(t2n(y1), ref, rtol=RTOL, atol=ATOL)
    assert float((y1 - y0).abs().max()) > 1e-6                # the statistics did change the eval-mode output


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_launches_follow_the_tensors_device():
    """The reference selects the GPU by argument (`create_model(dev=)`, `integrate(dev=)`): kernels must run on the device
    that holds the tensors, whatever the current device is."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import PWQuadManager
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    torch.cuda.set_device(0)
    torch.manual_seed(2)
    NF = PWQuadManager(n_flow=2)
    NF.create_model(2, 4, [3] * 3, dev=1)
    assert next(NF._model.parameters()).device.index == 1
    sig, err = NF.integrate(integrands.Camel(), 3, 20000, 1)
    assert abs(float(sig) - 0.2323) < 0.05
    gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4)
    r = torch.rand(3000, 8, dtype=torch.double, device="cuda:1")
    mom, wgt = gen.generateKinematics_batch(1000., r)
    rm, rw = orambo.generate_kinematics([100.] * 2, [100.] * 4, 1000., t2n(r))
    np.testing.assert_allclose(t2n(wgt), rw, rtol=1e-9)
    assert torch.cuda.current_device() == 0


@pytest.mark.parametrize("kind,c,fname", [("pwlin", dict(d=8, P=4, n_cells=4, K=64, NN=[32, 32, 32], roll=4), "normal"),
                                          ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32, 32, 32]), "normal"),
                                          ("pwquad", dict(d=4, n_cells=4, K=8, NN=[16, 16]), "camel"),
                                          ("pwquad", dict(d=16, n_cells=8, K=16, NN=[32, 32, 32]), "one")])
def test_fused_integrand_rider_on_the_matrix_core_kernels(kind, c, fname):
    """integrate()'s loop body for wide conditioners: the built-in integrand * J and its moment records come out of the last
    cell's matrix-core kernel (no X / J planes written) and equal what the unfused route gives from the materialised XJ
    (oracle integrand on the HIP forward's output), for G = 1 and for batched iterations."""
    from nf_amd import integrands
    torch.manual_seed(c["d"] + c["K"])
    NF = build_manager(dict(kind=kind, **c))
    model = NF._model.train()
    eng = model.engine()
    assert eng.fusable()
    d = c["d"]
    f = {"normal": integrands.Normal(0.5, 0.3, d), "camel": integrands.Camel(), "one": integrands.One()}[fname]
    G, N = 3, 20011
    w = torch.rand(G, N, d, dtype=torch.float32, device=dev())
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = torch.zeros((G, 4), dtype=torch.double, device=dev())
    eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    sd_fused = {k: t2n(v) for k, v in model.state_dict().items()}
    for g in range(G):
        model.load_state_dict(sd0) if g == 0 else None
        XJ = t2n(model.flow_forward(w[g], want_xj=True)["xj"])
        X, J = XJ[:, :-1], XJ[:, -1]
        fx = {"normal": lambda: oflow.normal(X, 0.5, 0.3, d), "camel": lambda: oflow.camel(X), "one": lambda: np.ones(N)}[fname]()
        n, mean, var, mx = oflow.moments(fx * J)
        rec = t2n(out[g])
        assert rec[0] == n
        np.testing.assert_allclose(rec[1], mean, rtol=1e-11)
        np.testing.assert_allclose(rec[2] / (n - 1), var, rtol=1e-9)
        np.testing.assert_allclose(rec[3], mx, rtol=1e-10)
    for k, v in model.state_dict().items():                     # the G fused iterations advanced the running statistics like G forwards
        np.testing.assert_allclose(sd_fused[k], t2n(v), rtol=1e-9, atol=1e-14, err_msg=k)
    one = torch.zeros((1, 4), dtype=torch.double, device=dev())
    model.load_state_dict(sd0)
    eng.integrate_steps(w[0], True, f.nis_kind, f.nis_params, one)
    np.testing.assert_allclose(t2n(one[0]), t2n(out[0]), rtol=1e-12)


# ------------------------------------------------------------------------------ minibatches of one optimizer step as grid.y
def _batched_models():
    from nf_amd.normalizing_flows.manager import AffineManager, PWLinManager, PWQuadManager
    out = []

    def add(name, mk):
        out.append(pytest.param(mk, id=name))

    def quad(d, nc, K, NN):
        def mk():
            NF = PWQuadManager(n_flow=d)
            NF.create_model(nc, K, NN)
            return NF
        return mk

    def lin(d, P, nc, K, NN, roll):
        def mk():
            NF = PWLinManager(n_flow=d)
            NF.create_model(P, nc, K, NN, roll)
            return NF
        return mk

    def aff(d, P, nc, NN):
        def mk():
            NF = AffineManager(n_flow=d)
            NF.create_model(P, nc, NN, 2)
            return NF
        return mk

    add("c2-class", quad(2, 2, 4, [3, 3, 3]))               # last-layer gradients ride in the head kernel
    add("d2-k8-w4", quad(2, 2, 8, [4, 4]))                   # narrow, separate last-layer reduction
    add("d5-k6-w8", quad(5, 4, 6, [7, 8, 5]))                # WM = 8, several transformed columns
    add("pwlin-d4", lin(4, 2, 3, 24, [6, 6], 2))
    add("affine-d4", aff(4, 2, 3, [5, 5]))
    add("wide-d6", quad(6, 3, 8, [16, 16]))                  # matrix-core adjoint: one minibatch at a time, blobs added up
    return out


@pytest.mark.parametrize("mk", _batched_models())
@pytest.mark.parametrize("mode", ["var", "est", "preburn"])
def test_batched_minibatches_equal_single_minibatch_calls(mk, mode):
    """nis_flow_forward_hist_batched / nis_loss_terms_batched / nis_flow_backward_loss_batched (G minibatches as the y dimension of
    every launch) against G calls of the per-minibatch entry points (pinned to the reference by the golden training runs): history
    planes, saved BatchNorm records, loss bookkeeping rows, running statistics after the call, and the summed parameter gradients."""
    import copy
    from nf_amd import integrands
    torch.manual_seed(11)
    NF = mk()
    model = NF._model.to(dev()).train()
    d = NF.n_flow
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.2 * torch.randn_like(p))
    G, N = 3, 1537
    w = torch.rand(G, N, d, dtype=torch.double, device=dev()) * (1 - 1e-9)
    f = integrands.Camel() if d == 2 else integrands.Normal(0.5, 0.35, d)
    maxf = torch.tensor(1.7, dtype=torch.double, device=dev())
    gloss = torch.tensor(1.0 / G, dtype=torch.double, device=dev())
    preburn, est = mode == "preburn", mode == "est"
    model_b = copy.deepcopy(model)
    ea, eb = model.engine(), model_b.engine()
    assert ea.backward_supported()
    # per minibatch
    hist_a, bn_a, out_a, g_a = [], [], [], None
    for g in range(G):
        xj, saved = ea.forward_train(w[g], False)
        fv, rec, out = ea.loss_terms(xj, maxf, preburn, est, f=f, w=w[g])
        gf = ea.backward_loss(saved, fv, maxf, rec, gloss, est)
        hist_a.append(saved[0])
        bn_a.append(saved[1])
        out_a.append(out)
        g_a = [x.clone() for x in gf] if g_a is None else [a + x for a, x in zip(g_a, gf)]
    # batched
    _, saved = eb.forward_train_batched(w, True)
    fv, rec, out = eb.loss_terms_batched(saved, maxf, preburn, est, f=f, w=w)
    g_b = eb.backward_loss_batched(saved, fv, maxf, rec, gloss, est)
    hb, ha = saved[0].clone(), torch.stack(hist_a)
    hb[:, 0, d], ha[:, 0, d] = 0.0, 0.0            # the input set's Jacobian plane is implied (AddJacobian: ones), never written
    np.testing.assert_allclose(t2n(hb), t2n(ha), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(t2n(saved[1]), t2n(torch.stack(bn_a)), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(t2n(out)[:, :6], t2n(torch.stack(out_a))[:, :6], rtol=1e-11, atol=1e-14)
    for (ka, va), (kb, vb) in zip(model.state_dict().items(), model_b.state_dict().items()):
        np.testing.assert_allclose(t2n(vb), t2n(va), rtol=1e-12, atol=1e-14, err_msg=ka)      # running statistics: minibatch order
    for ci, (a, b) in enumerate(zip(g_a, g_b)):
        scale = float(a.abs().max())
        np.testing.assert_allclose(t2n(b), t2n(a), rtol=1e-9, atol=1e-12 * max(scale, 1.0), err_msg=f"cell {ci}")


@pytest.mark.parametrize("which", ["c2-class", "wide-d6"])
def test_many_small_minibatches_in_one_call(which):
    """70 minibatches of 130 samples as the y / z extent of the training launches (every minibatch gets ONE workgroup per launch) against
    one call per minibatch: bookkeeping rows and the summed gradients."""
    from nf_amd import integrands
    import copy
    mk = [p.values[0] for p in _batched_models() if p.id == which][0]
    torch.manual_seed(2)
    NF = mk()
    model = NF._model.to(dev()).train()
    d = NF.n_flow
    G, N = 70, 130
    w = torch.rand(G, N, d, dtype=torch.double, device=dev()) * (1 - 1e-9)
    f = integrands.Camel() if d == 2 else integrands.Normal(0.5, 0.35, d)
    maxf = torch.tensor(1.3, dtype=torch.double, device=dev())
    gloss = torch.tensor(1.0 / G, dtype=torch.double, device=dev())
    model_b = copy.deepcopy(model)
    ea, eb = model.engine(), model_b.engine()
    out_a, g_a = [], None
    for g in range(G):
        xj, saved = ea.forward_train(w[g], False)
        fv, rec, out = ea.loss_terms(xj, maxf, False, False, f=f, w=w[g])
        gf = ea.backward_loss(saved, fv, maxf, rec, gloss, False)
        out_a.append(out)
        g_a = [x.clone() for x in gf] if g_a is None else [a + x for a, x in zip(g_a, gf)]
    _, saved = eb.forward_train_batched(w, True)
    fv, rec, out = eb.loss_terms_batched(saved, maxf, False, False, f=f, w=w)
    g_b = eb.backward_loss_batched(saved, fv, maxf, rec, gloss, False)
    np.testing.assert_allclose(t2n(out)[:, :6], t2n(torch.stack(out_a))[:, :6], rtol=1e-11, atol=1e-14)
    for ci, (a, b) in enumerate(zip(g_a, g_b)):
        np.testing.assert_allclose(t2n(b), t2n(a), rtol=1e-9, atol=1e-12 * max(float(a.abs().max()), 1.0), err_msg=f"cell {ci}")
    for (ka, va), (kb, vb) in zip(model.state_dict().items(), model_b.state_dict().items()):
        np.testing.assert_allclose(t2n(vb), t2n(va), rtol=1e-12, atol=1e-14, err_msg=ka)


def test_batched_minibatches_with_a_callable_integrand_and_xj():
    """The callable-integrand form of the fused epoch: xj (G, N, d+1) of the batched forward equals the per-minibatch outputs."""
    from nf_amd.normalizing_flows.manager import PWQuadManager
    torch.manual_seed(4)
    NF = PWQuadManager(n_flow=3)
    NF.create_model(3, 5, [6, 6])
    model = NF._model.to(dev()).train()
    w = torch.rand(4, 700, 3, dtype=torch.double, device=dev())
    import copy
    mb = copy.deepcopy(model)
    xj_b, _ = mb.engine().forward_train_batched(w, True, want_xj=True)
    for g in range(4):
        xj, _ = model.engine().forward_train(w[g], False)
        np.testing.assert_allclose(t2n(xj_b[g]), t2n(xj), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("tag", ["C4", "C5"])
def test_bench_topology_flow_into_rambo_moments_at_2e5_vs_oracle(tag, g_wide):
    """The C4 / C5 bench legs as a COMPOSITION at N = 200 000: width-32 matrix-core flow (the models of wide.npz, whose forward the
    reference pins at this N) with train-mode BatchNorm -> RAMBO weight (2 -> 4 massive / 2 -> 6 massless with pT / dR / eta cuts,
    tau-y sampling) * J -> fused moment record (`nis_rambo_moments`: no momenta, no weights, no X materialised), against the
    oracle's flow forward + oracle's phase-space weight on the same float32-lattice draws."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    case = f"fwd_{tag}"
    m = g_wide.meta[case]
    d, N = m["d"], m["N"]
    params = g_wide.sub(case, "sd")
    NF = build_manager(m, params)
    NF.best_model = NF._model
    NF._model.train()
    if tag == "C4":
        init, final, E, cuts, pdgs = [0., 0.], [100.] * 4, 13000., (0, 0, -1), [2, -1]
    else:
        init, final, E, cuts, pdgs = [0., 0.], [0.] * 6, 13000., (20, 0.4, 2.5), [21, 21]
    gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=True)
    f = integrands.RamboWeight(gen, E, *cuts, pdgs)
    w = kronecker_w(N, d).astype(np.float32)
    w = np.minimum(w, np.float32(1.0 - 2.0 ** -24))          # float32 rounding must not produce 1.0 (uniform_ never does)
    steps = oflow.plan_pwquad(d, m["n_cells"], m["K"])
    XJ = oflow.model_forward(w.astype(np.float64), {k: np.asarray(v) for k, v in params.items()}, steps, True)
    _, wref = orambo.generate_kinematics(init, final, E, XJ[:, :d], *cuts, pdf_active=True, tau=True)
    v = wref * XJ[:, d]
    NF.sample_hook = Replay([w])
    got = t2n(NF.moments_batches(f, 1, N, dev()))[0]
    assert got[0] == N
    np.testing.assert_allclose(got[1], v.mean(), rtol=1e-8)
    np.testing.assert_allclose(got[2], ((v - v.mean()) ** 2).sum(), rtol=1e-7)
    np.testing.assert_allclose(got[3], v.max(), rtol=1e-8)
    assert (v > 0).mean() > 0.01                              # the cuts leave a populated sample


@pytest.mark.parametrize("shape", [dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]), dict(kind="pwquad", d=5, n_cells=3, K=6, NN=[8, 8]),
                                   dict(kind="pwlin", d=4, P=2, n_cells=3, K=16, NN=[16, 16], roll=2)])
def test_integrate_with_a_python_callable_batched_flow_equals_one_forward_per_iteration(shape):
    """integrate(f, ...) with an arbitrary callable: the flow part of G iterations as ONE launch set that keeps X / J in the column planes
    (NIS_F_NONE) against the per-iteration path (one forward + materialised XJ per iteration) on the same draws: moment records,
    the combined estimate and the running statistics left behind."""
    import copy
    torch.manual_seed(21)
    A = build_manager(shape)
    A.best_model = A._model
    A._model.train()
    B = build_manager(shape)
    B._model.load_state_dict(copy.deepcopy(A._model.state_dict()))
    B.best_model = B._model
    B._model.train()
    d, N, nitn = shape["d"], 30_011, 7

    def f(x):
        assert x.dim() == 2 and x.shape[1] == d and x.dtype == torch.double
        return torch.exp(-((x - 0.45) ** 2).sum(-1) * 4.0) + 0.05 * x[:, 0]

    g = torch.Generator().manual_seed(3)
    draws = [torch.rand(N, d, generator=g, dtype=torch.float32) * (1 - 2 ** -24) for _ in range(nitn)]
    A.batch_iterations = 3                      # sets of 3, 3, 1
    A.sample_hook = Replay([x.numpy() for x in draws])
    B.batch_callable = False
    B.sample_hook = Replay([x.numpy() for x in draws])
    ma = t2n(A.moments_batches(f, nitn, N, dev()))
    mb = t2n(B.moments_batches(f, nitn, N, dev()))
    np.testing.assert_allclose(ma, mb, rtol=1e-11)
    for (ka, va), (kb, vb) in zip(A._model.state_dict().items(), B._model.state_dict().items()):
        np.testing.assert_allclose(t2n(va), t2n(vb), rtol=1e-12, atol=1e-15, err_msg=ka)
    A.sample_hook = None                        # and with the draw inside the launch set
    sig, err = A.integrate(f, 4, 50_000, 0)
    assert np.isfinite(float(sig)) and float(err) > 0


def test_build_then_smoke_in_one_process():
    """The driver's two hooks in ONE interpreter, build() first: the library is opened before anything imported torch there (it used to
    bind to a second HIP runtime then, and smoke()'s first launch failed)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('BUILD_SMOKE_OK')"], capture_output=True,
                       text=True, cwd=root, timeout=900)
    assert r.returncode == 0 and "BUILD_SMOKE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_no_access_past_the_end_of_caller_buffers():
    """tools/guard_check.py: inputs, kept planes and outputs of the main entry points placed at the very end of their own allocator
    segment, small and ragged batch sizes (a prefetch past a batch of fewer than 64 samples once read off the end of `w`).  A memory
    fault aborts the process, hence the subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for which in ("1", "2"):          # 2: the buffers the Python layer allocates sit at segment ends as well
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "guard_check.py"), which], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "GUARD_OK" in r.stdout, which + r.stdout[-1500:] + r.stderr[-2500:]


@pytest.mark.parametrize("shape", [dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3]), dict(kind="pwquad", d=5, n_cells=3, K=6, NN=[16, 16])])
def test_captured_training_step_equals_eager(shape, monkeypatch):
    """flow.TrainPlan (opt-in, NIS_TRAIN_GRAPH=1): forward + loss bookkeeping and the adjoint of an optimizer step as two hipGraph
    replays against the same library calls launched eagerly -- loss history, estimator bookkeeping, gradients, parameters and running
    statistics after a few epochs that cross the preburn switch (two plans), identical draws."""
    import copy
    from nf_amd import integrands
    d = shape["d"]
    torch.manual_seed(31)
    base = build_manager(shape)
    f = integrands.Camel() if d == 2 else integrands.Normal(0.5, 0.3, d)
    g = torch.Generator().manual_seed(8)
    draws = [torch.rand(1500, d, dtype=torch.double, generator=g) for _ in range(80)]

    def run(graph):
        if graph:
            monkeypatch.setenv("NIS_TRAIN_GRAPH", "1")
        else:
            monkeypatch.delenv("NIS_TRAIN_GRAPH", raising=False)
        NF = build_manager(shape)
        NF._model = copy.deepcopy(base._model)
        NF.best_model = NF._model
        it = iter(draws)
        NF.sample_hook = lambda n, dd, dtype, stage: next(it)[:n].to(dtype)
        optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        res = NF._train_variance_forward_seq(f, optim, False, None, 3000, 8, 0, True, True, mini_batch_size=1000, integrate=True, preburn_time=3)
        eng = NF._model.engine()
        assert bool(getattr(eng, "_train_plans", None)) == graph
        if graph:
            assert all(p.captured for p in eng._train_plans.values())
        return res, t2n(torch.stack(NF.history)), [t2n(p.grad) for p in NF._model.parameters()], {k: t2n(v) for k, v in NF._model.state_dict().items()}

    ra, ha, ga, sa = run(True)
    rb, hb, gb, sb = run(False)
    np.testing.assert_allclose(ha, hb, rtol=1e-12)
    np.testing.assert_allclose(ra, rb, rtol=1e-9)
    for x, y in zip(ga, gb):
        np.testing.assert_allclose(x, y, rtol=1e-11, atol=1e-16)
    for k in sa:
        np.testing.assert_allclose(sa[k], sb[k], rtol=1e-11, atol=1e-15, err_msg=k)


def test_captured_wide_training_step_with_the_side_stream_equals_eager(monkeypatch):
    """Matrix-core adjoint at a batch that takes the side stream (flow_bwd_wm.hip, bwd_wide_run: the last layer's weight-gradient GEMM
    runs beside the layer chain once N x G >= 65 536): the fork / join must survive a stream capture (flow.TrainPlan) and give the
    gradients of the eager launches, which in turn equal the one-stream path of a batch below the threshold (checked against the oracle
    elsewhere); two epochs of two minibatches of 33 000."""
    import copy
    from nf_amd import integrands
    shape = dict(kind="pwquad", d=4, n_cells=2, K=8, NN=[16, 16])
    torch.manual_seed(33)
    base = build_manager(shape)
    f = integrands.Normal(0.5, 0.3, 4)
    g = torch.Generator().manual_seed(10)
    draws = [torch.rand(33000, 4, dtype=torch.double, generator=g) for _ in range(16)]

    def run(graph):
        if graph:
            monkeypatch.setenv("NIS_TRAIN_GRAPH", "1")
        else:
            monkeypatch.delenv("NIS_TRAIN_GRAPH", raising=False)
        NF = build_manager(shape)
        NF._model = copy.deepcopy(base._model)
        NF.best_model = NF._model
        it = iter(draws)
        NF.sample_hook = lambda n, dd, dtype, stage: next(it)[:n].to(dtype)
        optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        NF._train_variance_forward_seq(f, optim, False, None, 66000, 2, 0, True, True, mini_batch_size=33000, integrate=False, preburn_time=0)
        eng = NF._model.engine()
        assert bool(getattr(eng, "_train_plans", None)) == graph
        return t2n(torch.stack(NF.history)), [t2n(p.grad) for p in NF._model.parameters()]

    ha, ga = run(True)
    hb, gb = run(False)
    np.testing.assert_allclose(ha, hb, rtol=1e-12)
    for x, y in zip(ga, gb):
        np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-16)


def test_captured_training_step_with_more_chunks_than_one_plan_holds(monkeypatch):
    """a TrainPlan owns ONE history.  An epoch of more minibatches than one fused call carries runs every chunk's
    forward before the single loss.backward() (manager.py:212-280); the second chunk of equal shape must NOT reuse the plan while the
    first one's backward is outstanding (it used to: both backward nodes then replayed on the LAST chunk's history -- silently
    2 x grad(chunk 2)).  Now the busy plan sends the second chunk down the eager path; gradients equal the all-eager run, and a direct
    misuse of the plan raises."""
    import copy
    from nf_amd import _lib, integrands
    from nf_amd.normalizing_flows import manager as mgr
    shape = dict(kind="pwquad", d=2, n_cells=2, K=4, NN=[3, 3, 3])
    torch.manual_seed(32)
    base = build_manager(shape)
    f = integrands.Camel()
    g = torch.Generator().manual_seed(9)
    draws = [torch.rand(1000, 2, dtype=torch.double, generator=g) for _ in range(40)]
    monkeypatch.setattr(mgr.BasicManager, "_minibatches_per_call", lambda self, mbs: 2)     # 4 minibatches per epoch = two chunks of 2

    def run(graph):
        if graph:
            monkeypatch.setenv("NIS_TRAIN_GRAPH", "1")
        else:
            monkeypatch.delenv("NIS_TRAIN_GRAPH", raising=False)
        NF = build_manager(shape)
        NF._model = copy.deepcopy(base._model)
        NF.best_model = NF._model
        it = iter(draws)
        NF.sample_hook = lambda n, dd, dtype, stage: next(it)[:n].to(dtype)
        optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        NF._train_variance_forward_seq(f, optim, False, None, 4000, 3, 0, True, True, mini_batch_size=1000, integrate=False, preburn_time=0)
        return NF, t2n(torch.stack(NF.history)), [t2n(p.grad) for p in NF._model.parameters()]

    NFa, ha, ga = run(True)
    plans = NFa._model.engine()._train_plans
    assert plans and all(p.captured and not p.busy() for p in plans.values())
    _, hb, gb = run(False)
    np.testing.assert_allclose(ha, hb, rtol=1e-12)
    for x, y in zip(ga, gb):
        np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-16)
    # direct misuse: a second forward while a holder of the first is alive, and a backward of an overwritten generation
    plan = next(iter(plans.values()))

    class Tok:
        pass
    t1 = Tok()
    w = torch.rand(plan.G, plan.N, 2, dtype=torch.double, device=dev())
    plan.forward(w, holder=t1)
    gen1 = plan.generation
    assert plan.busy()
    with pytest.raises(_lib.NisError, match="awaiting its backward"):
        plan.forward(w, holder=Tok())
    del t1
    assert not plan.busy()
    plan.forward(w)
    with pytest.raises(_lib.NisError, match="stale history"):
        plan.backward(torch.ones(1, dtype=torch.double, device=dev()), gen1)


# NIS_TEST_SWEEP_SEED=<int> shifts the seed of every random sweep below (other shapes, same tests); NIS_TEST_SWEEP_SCALE=<k> multiplies the
# number of cases.  The default run is the committed, fixed set.  Shifted seeds can hit knife-edge samples that are ambiguous in the
# reference itself (a width at the threshold of vanishing in the cumulative sum makes the bin depend on the last bit of exp();
# a phase-space point at threshold) -- seen in 2 of 40 seeds x 366 cases, 2 elements each.
_SWEEP_SEED = int(os.environ.get("NIS_TEST_SWEEP_SEED", "0"))
_SWEEP_SCALE = max(1, int(os.environ.get("NIS_TEST_SWEEP_SCALE", "1")))


def _random_cell_shapes(n, seed):
    rng = np.random.default_rng(seed + _SWEEP_SEED)
    n *= _SWEEP_SCALE
    out = []
    for i in range(n):
        kind = "pwquad" if rng.random() < 0.6 else "pwlin"
        d = int(rng.integers(2, 13))
        P = int(rng.integers(1, d))
        K = int(rng.choice([2, 3, 4, 5, 8, 13, 16, 31, 32, 33, 47, 64, 100, 128]))
        depth = int(rng.integers(1, 4))
        wmax = int(rng.choice([3, 8, 16, 32, 64]))
        NN = [int(rng.integers(1, wmax + 1)) for _ in range(depth)]
        N = int(rng.integers(17, 700))
        out.append(pytest.param(kind, d, P, K, NN, N, id=f"{i}-{kind}-d{d}-P{P}-K{K}-NN{'x'.join(map(str, NN))}-N{N}"))
    return out


@pytest.mark.parametrize("kind,d,P,K,NN,N", _random_cell_shapes(28, 20261004))
def test_random_cell_shapes_forward_and_adjoint(kind, d, P, K, NN, N):
    """A seeded sweep over cell shapes nobody picked by hand (bin counts 2 .. 128, widths 1 .. 64, depths 1 .. 3, every pass-through /
    transformed split of d <= 12, ragged N): train- and eval-mode forward against the oracle (bins bit-exact, values 1e-9), and the
    adjoint against autograd through the stock-PyTorch re-evaluation of the same cell."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    torch.manual_seed(1000 * d + K)
    cell = (PWQuad if kind == "pwquad" else PWLin)(d, P, K, list(NN)).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.25 * torch.randn_like(p))
        for mod in cell.NN:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(0, 0.5)
                mod.running_var.uniform_(0.1, 1.0)
    params = {"c." + k: t2n(v) for k, v in cell.state_dict().items()}
    x = torch.rand(N, d + 1, dtype=torch.double)
    x[:, -1] += 0.5
    x[:, :d] *= (1 - 1e-12)
    cell = cell.to(dev())
    for train in (True, False):
        want, upd = {}, {}
        y_ref = oflow.cell_forward(x.numpy(), params, "c", kind, P, K, train, upd, want)
        cell.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in params.items()})
        cell.train(train)
        res = cell._own_engine().forward(x.to(dev()), True, train, want_xj=True, want_bins=True)
        np.testing.assert_array_equal(t2n(res["bins"]).T, want["bins"])
        np.testing.assert_allclose(t2n(res["xj"]), y_ref, rtol=RTOL, atol=ATOL)
    # adjoint (train mode), same cell through the model-level autograd function
    cell.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in params.items()})
    cell.train(True)
    model = torch.nn.Sequential(cell)
    R = torch.randn(N, d + 1, dtype=torch.double, device=dev())

    keep = None

    def grads(force_torch):
        for p in cell.parameters():
            p.grad = None
        xx = x.to(dev()).clone().requires_grad_(True)
        out = model_torch(model, xx) if force_torch else cell(xx)
        out = torch.where(keep[:, None], out, torch.zeros_like(out))
        (out * R).sum().backward()
        return xx.grad.clone(), [p.grad.clone() for p in cell.parameters()]

    sd = {k: v.clone() for k, v in cell.state_dict().items()}
    for train in (True, False):                       # the reference's autograd differentiates a model in any mode
        cell.train(train)
        model.train(train)
        # The stock-PyTorch re-evaluation runs on the GPU, where torch.cumsum is a parallel scan: its bin edges differ from the sequential
        # sums of the reference on the CPU (= the oracle = the kernels, asserted above) in the last bits, and where widths vanish in the sum
        # (many bins, wide random nets) a sample can sit on an edge and take ANOTHER bin there.  The adjoint is compared on the samples whose
        # forward values agree, i.e. where both differentiate the same branch; the others are masked out of the loss on both sides.
        with torch.no_grad():
            cell.load_state_dict(sd)
            o_h = cell(x.to(dev()))
            cell.load_state_dict(sd)
            o_t = model_torch(model, x.to(dev()))
        keep = ((o_h - o_t).abs() <= 1e-8 * o_t.abs() + 1e-12).all(dim=1)
        # a handful of edge samples, not a tenth of the batch: a forward bug that touches more rows than that fails here
        assert int((~keep).sum()) <= max(3, N // 200), f"train={train}: {int((~keep).sum())} of {N} samples disagree in the forward pass"
        cell.load_state_dict(sd)
        gx_h, gp_h = grads(False)
        cell.load_state_dict(sd)
        gx_t, gp_t = grads(True)
        # A sample that lands in a bin of (nearly) vanishing width has derivative terms of 1e15 .. 1e25 that cancel to O(1): the two
        # evaluation orders then differ by cond * eps.  Rows are independent samples, so the input gradient is compared per row against
        # that row's own scale; such a cell's parameter gradients (sums over the rows) get the looser factor as a whole.
        ght, gtt = t2n(gx_h), t2n(gx_t)
        fac = 1e-7 if float(np.abs(gtt).max()) > 1e8 else 1e-9
        rowscale = np.maximum(1.0, np.abs(gtt).max(axis=1, keepdims=True))
        atol = 1e-7 * rowscale if fac == 1e-7 else 1e-9 * max(1.0, float(np.abs(gtt).max()))
        bad = np.abs(ght - gtt) > 1e-6 * np.abs(gtt) + atol
        assert not bad.any(), f"train={train}: {int(bad.sum())} input-gradient entries differ, max |diff| {float(np.abs(ght - gtt)[bad].max()):.3e}"
        gpmax = max(1.0, max(float(b.abs().max()) for b in gp_t))
        for (name, _), a, b in zip(cell.named_parameters(), gp_h, gp_t):
            np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-6, atol=fac * (gpmax if fac == 1e-7 else max(1.0, float(b.abs().max()))),
                                       err_msg=f"{name} train={train}")


def test_knife_edge_cell_bins_equal_the_oracle_wherever_the_ambiguity_flag_is_clear():
    """The contract behind "bit-exact bins" (VERDICT r3 item 5), on the cell a shifted sweep seed found (NIS_TEST_SWEEP_SEED=39, case 56: a
    47-bin PWQuad cell behind a 45 x 63 x 60 net whose widths span 1e+-17): the reference's argmax compares cumsum(W)/sum(W) with x and
    takes the FIRST maximum (coupling_cells.py:198-202); where a width sits on the threshold of vanishing in the running sum, whether two
    edges are equal turns on the last bit of exp() -- the reference's own CPU and GPU builds disagree there, and so may this library
    (nis_exp is within 1-2 ulp of libm, not bit-identical).  The kernels now SAY so: bins_out carries NIS_BIN_AMBIGUOUS_FLAG (and the
    status words NIS_STATUS_BIN_AMBIGUOUS) for exactly those decisions -- an edge quotient within 2^-50 of x, or a width between 2^-55 and
    2^-51 of the running sum at the end of the kept prefix.  Asserted: every bin WITHOUT the flag equals the oracle's; the flagged set is
    small even for this cell (<= 2.5 % of its decisions: its widths are spread over 34 decades, so a few dozen land in the 16-fold
    threshold window) and contains every disagreement; values of unflagged samples agree to 1e-9."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWQuad
    d, P, K, NN, N = 9, 4, 47, [45, 63, 60], 508
    torch.manual_seed(1000 * d + K)                    # the sweep's construction, seed and all (test_random_cell_shapes_forward_and_adjoint)
    cell = PWQuad(d, P, K, list(NN)).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.25 * torch.randn_like(p))
        for mod in cell.NN:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(0, 0.5)
                mod.running_var.uniform_(0.1, 1.0)
    params = {"c." + k: t2n(v) for k, v in cell.state_dict().items()}
    x = torch.rand(N, d + 1, dtype=torch.double)
    x[:, -1] += 0.5
    x[:, :d] *= (1 - 1e-12)
    cell = cell.to(dev())
    seen_flag = 0
    for train in (True, False):
        want, upd = {}, {}
        y_ref = oflow.cell_forward(x.numpy(), params, "c", "pwquad", P, K, train, upd, want)
        cell.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in params.items()})
        cell.train(train)
        eng = cell._own_engine()
        eng.clear_bin_ambiguous()
        res = eng.forward(x.to(dev()), True, train, want_xj=True, want_bins=True)
        bins, amb = t2n(res["bins"]).T, t2n(res["ambiguous"]).T
        differ = bins != want["bins"]
        assert not (differ & ~amb).any(), f"train={train}: {int((differ & ~amb).sum())} unflagged bins differ from the oracle"
        assert amb.sum() <= 0.025 * amb.size, f"train={train}: {int(amb.sum())} of {amb.size} decisions flagged"
        assert bool(eng.bin_ambiguous()) == bool(amb.any())
        clear_rows = ~amb.any(axis=1)
        np.testing.assert_allclose(t2n(res["xj"])[clear_rows], y_ref[clear_rows], rtol=RTOL, atol=ATOL)
        seen_flag += int(amb.sum())
    assert seen_flag >= 1, "this cell is the one with threshold widths: the flag must fire on it"


def test_knife_edge_phase_space_point_stated_tolerance():
    """The other knife-edge sample of the shifted sweeps (NIS_TEST_SWEEP_SEED=27, case 33): a 2 -> 3 map with massive beams and one massive
    leg, 8 385 points, in which TWO of 167 700 momentum components differ from the oracle by 6.7e-8 relative (everything else: 1e-8 and
    better).  Not the solver: the kernel's monotone Newton iteration and its fixed-depth lattice bisection (`bisect_levels=180`, the depth
    the reference's own bisection reaches for a batch of this size, flat_phase_space_generator.py:327-357) agree with EACH OTHER to 1e-10
    on this batch (asserted), so the two components come from the decay / boost algebra the kernel evaluates in another order than the
    reference (on-shell boost form, shared rho numerators, reciprocals) at a badly conditioned point.  Tolerances
    stated: every component 1e-7 of the event scale, all but a handful 1e-8, weights 1e-7, identical cut decisions."""
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    init, final = [33.258424188333784, 7.969856530883729], [0.0, 0.0, 138.76266686521828]
    E, cuts, N = 11705.02016708497, (8.481629858856799, 0.574695152402175, 2.2981017231981884), 8385
    r = np.random.default_rng(N).random((N, 5))
    ref_m, ref_w = orambo.generate_kinematics(init, final, E, r, *cuts, pdf_active=False, tau=False)
    rt = torch.as_tensor(r).to(dev())
    gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=False, tau=False)
    mom, wgt = gen.generateKinematics_batch(E, rt, *cuts)
    err = np.abs(t2n(mom) - ref_m) / (np.abs(ref_m) + 1e-9 * E)
    assert err.max() <= 1e-7, err.max()
    assert (err > 1e-8).sum() <= 8, int((err > 1e-8).sum())
    same_cut = (t2n(wgt) == 0) == (ref_w == 0)
    assert (~same_cut).sum() <= 2
    np.testing.assert_allclose(t2n(wgt)[same_cut], ref_w[same_cut], rtol=1e-7)
    gen_l = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=False, tau=False)
    gen_l.bisect_levels = 180
    mom_l, wgt_l = gen_l.generateKinematics_batch(E, rt, *cuts)
    np.testing.assert_allclose(t2n(mom_l), t2n(mom), rtol=1e-10, atol=1e-10 * E)


def _random_rambo_shapes(n_cases, seed):
    rng = np.random.default_rng(seed + _SWEEP_SEED)
    n_cases *= _SWEEP_SCALE
    out = []
    for i in range(n_cases):
        n = int(rng.integers(2, 11))
        style = int(rng.integers(0, 3))            # 0 massless, 1 massive, 2 mixed
        final = [0.0 if style == 0 or (style == 2 and rng.random() < 0.5) else float(rng.uniform(1, 150)) for _ in range(n)]
        pdf = bool(rng.random() < 0.5)
        tau = bool(rng.random() < 0.7)
        init = [0., 0.] if pdf or rng.random() < 0.5 else [float(rng.uniform(1, 50)), float(rng.uniform(1, 50))]
        cuts = (-1, -1, -1) if rng.random() < 0.4 else (float(rng.uniform(1, 30)), float(rng.uniform(0.1, 0.6)), float(rng.uniform(2, 5)))
        E = float(rng.uniform(2000, 13000))
        N = int(rng.integers(100, 9000))
        out.append(pytest.param(init, final, E, cuts, pdf, tau, N, id=f"{i}-n{n}-{'mass0 massive mixed'.split()[style]}-pdf{int(pdf)}-tau{int(tau)}-cuts{int(cuts[0] > 0)}-N{N}"))
    return out


@pytest.mark.parametrize("init,final,E,cuts,pdf,tau,N", _random_rambo_shapes(16, 77))
def test_random_rambo_configurations_vs_oracle(init, final, E, cuts, pdf, tau, N):
    """A seeded sweep over 2 -> n maps nobody picked by hand (n = 2 .. 10, massless / massive / mixed legs, massive or massless beams, tau-y or
    x1-x2 sampling, with and without cuts): momenta and weights against the oracle, row input and column-plane input bit-identical."""
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    n = len(final)
    rng = np.random.default_rng(N)
    r = rng.random((N, 3 * n - 4 + (2 if pdf else 0)))
    ref_m, ref_w = orambo.generate_kinematics(init, final, E, r, *cuts, pdf_active=pdf, tau=tau)
    gen = FlatInvertiblePhasespace(init, final, pdf=None, pdf_active=pdf, tau=tau)
    rt = torch.as_tensor(r).to(dev())
    mom, wgt = gen.generateKinematics_batch(E, rt, *cuts)
    differ = (t2n(wgt) == 0) != (ref_w == 0)          # a point within rounding of a cut threshold may fall on either side
    assert differ.sum() <= 2
    ok = ~differ
    np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-8, atol=1e-9 * E)
    np.testing.assert_allclose(t2n(wgt)[ok], ref_w[ok], rtol=1e-8)
    planes = rt.t().contiguous()
    wp, _ = gen.weights_from_planes(E, planes, list(range(r.shape[1])), N, *cuts)
    np.testing.assert_array_equal(t2n(wp), t2n(wgt))


def _random_models(n_cases, seed):
    rng = np.random.default_rng(seed + _SWEEP_SEED)
    n_cases *= _SWEEP_SCALE
    out = []
    for i in range(n_cases):
        which = rng.choice(["pwquad", "pwquad", "pwlin", "affine"])
        depth = int(rng.integers(1, 4))
        wmax = int(rng.choice([3, 3, 8, 16, 32]))
        NN = [int(rng.integers(2, wmax + 1)) for _ in range(depth)]
        if which == "pwquad":
            d = int(rng.integers(2, 11))
            cfg = dict(kind="pwquad", d=d, n_cells=int(rng.integers(1, 5)), K=int(rng.choice([2, 4, 5, 8, 16, 21, 40])), NN=NN)
        else:
            d = int(rng.integers(2, 9))
            cfg = dict(kind=which, d=d, P=int(rng.integers(1, d)), n_cells=int(rng.integers(1, 5)), K=int(rng.choice([3, 8, 32, 70])), NN=NN,
                       roll=int(rng.integers(1, d)))
        N = int(rng.integers(40, 2500))
        G = int(rng.choice([1, 2, 5]))
        out.append(pytest.param(cfg, N, G, id=f"{i}-{which}-d{cfg['d']}-c{cfg['n_cells']}-K{cfg['K']}-NN{'x'.join(map(str, NN))}-N{N}-G{G}"))
    return out


@pytest.mark.parametrize("cfg,N,G", _random_models(20, 4242))
def test_random_models_batched_integrate_step_vs_oracle(cfg, N, G):
    """A seeded sweep over whole models nobody picked by hand (the three managers, d = 2 .. 10, 1 .. 4 requested cells, bins 2 .. 70, widths
    2 .. 32, ragged N, G batches per launch set): integrate()'s loop body with the fused integrand -- train-mode BatchNorm, G batches as
    grid.y, running statistics applied in batch order -- against the oracle's forward + integrand on the same draws; then the same
    minibatches through the training step (grid.y minibatches) against one call per minibatch."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import AffineManager
    torch.manual_seed(7 * N + G)
    d = cfg["d"]
    if cfg["kind"] == "affine":
        NF = AffineManager(n_flow=d)
        NF.create_model(cfg["P"], cfg["n_cells"], list(cfg["NN"]), cfg["roll"])
        steps = oflow.plan_affine(d, cfg["P"], cfg["n_cells"], cfg["roll"])
    else:
        NF = build_manager(cfg)
        steps = plan_from_meta(cfg)
    with torch.no_grad():
        for p in NF._model.parameters():
            p.add_(0.15 * torch.randn_like(p))
    model = NF.best_model
    model.train()
    eng = model.engine()
    f = integrands.Normal(0.45, 0.3, d)
    cur = {k: t2n(v) for k, v in model.state_dict().items()}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    w = (torch.rand(G, N, d, device=dev()) * (1 - 2 ** -23)).to(torch.float32)
    if eng.fusable():
        out = torch.zeros(G, 4, dtype=torch.double, device=dev())
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out)
    else:           # no fused rider in the last cell (matrix-core PWQuad, bin count not 4 / 8 / 16 / 32): kept planes + integrand from the planes
        NF.batch_iterations = G
        NF.sample_hook = Replay([t2n(w[g]) for g in range(G)])
        out = NF.moments_batches(f, G, N, dev())
        NF.sample_hook = None
    got = t2n(out)
    for g in range(G):
        upd = {}
        XJ = oflow.model_forward(t2n(w[g]).astype(np.float64), cur, steps, True, upd)
        cur.update(upd)
        v = oflow.normal(XJ[:, :d], 0.45, 0.3, d) * XJ[:, d]
        ref = np.array([N, v.mean(), ((v - v.mean()) ** 2).sum(), v.max()])
        np.testing.assert_allclose(got[g], ref, rtol=5e-9, atol=1e-300, err_msg=f"g={g}")
    sd1 = model.state_dict()
    for k_, v_ in cur.items():
        if "running" in k_:
            np.testing.assert_allclose(t2n(sd1[k_]), v_, rtol=1e-9, atol=1e-13, err_msg=k_)
    # the same batches as the minibatches of one optimizer step
    if not eng.backward_supported():
        return
    import copy
    model.load_state_dict(sd0)
    model_b = copy.deepcopy(model)
    wd = w.double()
    maxf = torch.tensor(0.9, dtype=torch.double, device=dev())
    gloss = torch.tensor(1.0 / G, dtype=torch.double, device=dev())
    g_a = None
    for g in range(G):
        xj, saved = eng.forward_train(wd[g], False)
        fv, rec, o8 = eng.loss_terms(xj, maxf, False, False, f=f, w=wd[g])
        gf = eng.backward_loss(saved, fv, maxf, rec, gloss, False)
        g_a = [x.clone() for x in gf] if g_a is None else [a + x for a, x in zip(g_a, gf)]
    eb = model_b.engine()
    _, saved = eb.forward_train_batched(wd, True)
    fv, rec, o8 = eb.loss_terms_batched(saved, maxf, False, False, f=f, w=wd)
    g_b = eb.backward_loss_batched(saved, fv, maxf, rec, gloss, False)
    for ci, (a, b) in enumerate(zip(g_a, g_b)):
        np.testing.assert_allclose(t2n(b), t2n(a), rtol=1e-8, atol=1e-12 * max(float(a.abs().max()), 1e-30), err_msg=f"cell {ci}")


@pytest.mark.parametrize("cfg,N,G", _random_models(14, 99))
@pytest.mark.parametrize("train", [True, False])
def test_random_models_in_kernel_draws_equal_the_fed_path(cfg, N, G, train):
    """The same seeded model sweep through the draw-inside form of the launch set (`nis_integrate_steps_rng`) against the fed form on the
    materialised Philox rows (`nis_philox_rows`), train and eval mode, with the fused integrand where the model has the rider and with
    kept planes (NIS_F_NONE) everywhere: moment records, kept X / J planes and running statistics."""
    import copy
    from nf_amd import _lib, integrands
    from nf_amd.flow import philox_rows
    from nf_amd.normalizing_flows.manager import AffineManager
    torch.manual_seed(3 * N + G)
    d = cfg["d"]
    if cfg["kind"] == "affine":
        NF = AffineManager(n_flow=d)
        NF.create_model(cfg["P"], cfg["n_cells"], list(cfg["NN"]), cfg["roll"])
    else:
        NF = build_manager(cfg)
    with torch.no_grad():
        for p in NF._model.parameters():
            p.add_(0.15 * torch.randn_like(p))
    A = NF.best_model.train(train)
    B = copy.deepcopy(A)
    ea, eb = A.engine(), B.engine()
    f = integrands.Normal(0.45, 0.3, d)
    seed, s0 = 1234567 + N, 11
    rows = philox_rows(seed, s0, G, N, d, dev())
    pa = torch.zeros(G, d + 1, N, dtype=torch.double, device=dev())
    pb = torch.zeros(G, d + 1, N, dtype=torch.double, device=dev())
    ea.integrate_steps_rng(G, N, dev(), _lib.NIS_F_NONE, [], None, seed, s0, train=train, planes=pa, keep_planes=True)
    eb.integrate_steps(rows, train, _lib.NIS_F_NONE, [], None, planes=pb)
    np.testing.assert_allclose(t2n(pa), t2n(pb), rtol=1e-10 if train else 1e-13, atol=1e-300)      # train: BN_0 records summed about 1/2 vs about the wave's first value
    if ea.fusable():
        oa = torch.zeros(G, 4, dtype=torch.double, device=dev())
        ob = torch.zeros(G, 4, dtype=torch.double, device=dev())
        ea.integrate_steps_rng(G, N, dev(), f.nis_kind, f.nis_params, oa, seed, s0 + G, train=train)
        eb.integrate_steps(philox_rows(seed, s0 + G, G, N, d, dev()), train, f.nis_kind, f.nis_params, ob)
        np.testing.assert_allclose(t2n(oa), t2n(ob), rtol=1e-10)
    for (ka, va), (kb, vb) in zip(A.state_dict().items(), B.state_dict().items()):
        np.testing.assert_allclose(t2n(va), t2n(vb), rtol=1e-11, atol=1e-15, err_msg=ka)


@pytest.mark.parametrize("case", range(10))
def test_random_flow_into_rambo_fused_equals_materialised(case):
    """A seeded sweep over flow -> RAMBO pipelines (2 -> n with n = 2 .. 7, massless / massive legs, with and without cuts, narrow and
    matrix-core flows, G iterations per launch set): the fused form (kept planes -> `nis_rambo_moments`, nothing materialised) against
    the materialised form (X emitted, weights from `generateKinematics_batch`, `nis_moments`) on identical draws."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    rng = np.random.default_rng(500 + case)
    n = int(rng.integers(2, 8))
    pdf = bool(rng.random() < 0.6) or n == 2          # d = 3n - 4 (+ 2): at least 2 flow dimensions
    d = 3 * n - 4 + (2 if pdf else 0)
    massive = bool(rng.random() < 0.5)
    final = [float(rng.uniform(5, 120)) if massive else 0.0 for _ in range(n)]
    cuts = (-1, -1, -1) if rng.random() < 0.4 else (float(rng.uniform(5, 25)), float(rng.uniform(0.2, 0.5)), float(rng.uniform(2.5, 5)))
    E = float(rng.uniform(3000, 13000))
    NN = [int(rng.choice([3, 8, 16, 32]))] * int(rng.integers(1, 4))
    K = int(rng.choice([4, 6, 16]))
    N, G = int(rng.integers(300, 5000)), int(rng.choice([1, 3]))
    torch.manual_seed(case)
    cfg = dict(kind="pwquad", d=d, n_cells=2, K=K, NN=NN)
    NF = build_manager(cfg)
    params = {k: v.clone() for k, v in NF._model.state_dict().items()}
    gen = FlatInvertiblePhasespace([0., 0.], final, pdf=None, pdf_active=pdf, tau=True)
    f = integrands.RamboWeight(gen, E, *cuts)
    draws = [(torch.rand(N, d) * (1 - 2 ** -23)).to(torch.float32).numpy() for _ in range(G)]
    NF.batch_iterations = G
    NF.sample_hook = Replay(draws)
    fused = t2n(NF.moments_batches(f, G, N, dev()))
    NF2 = build_manager(cfg)
    NF2._model.load_state_dict(params)
    NF2.batch_callable = False
    NF2.sample_hook = Replay(draws)
    plain = t2n(NF2.moments_batches(lambda x: f(x), G, N, dev()))
    assert (fused[:, 0] == N).all()
    np.testing.assert_allclose(fused[:, 1], plain[:, 1], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(fused[:, 2], plain[:, 2], rtol=1e-8, atol=1e-300)
    np.testing.assert_allclose(fused[:, 3], plain[:, 3], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("cfg,N,G", _random_models(8, 31337))
@pytest.mark.parametrize("loss_mode", ["var", "est"])
def test_random_models_training_loop_fused_equals_piecewise(cfg, N, G, loss_mode):
    """The seeded model sweep through `_train_variance_forward_seq` itself: the fused epoch (grid.y minibatches, library-side loss and
    adjoint) against the piecewise path through torch's autograd graph -- loss history, returned estimate, gradients -- for a few
    epochs that cross the preburn switch."""
    import copy
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import AffineManager
    d = cfg["d"]
    torch.manual_seed(N + 17 * G)
    if cfg["kind"] == "affine":
        base = AffineManager(n_flow=d)
        base.create_model(cfg["P"], cfg["n_cells"], list(cfg["NN"]), cfg["roll"])
    else:
        base = build_manager(cfg)
    if not base._model.engine().backward_supported():
        pytest.skip("no adjoint for this shape")
    f = integrands.Normal(0.5, 0.3, d)
    mbs = max(64, N // 2)
    n_mb = max(1, G)
    gen = torch.Generator().manual_seed(N)
    draws = [torch.rand(2 * mbs, d, dtype=torch.double, generator=gen) * (1 - 1e-9) for _ in range(d + 6 * n_mb + 8)]

    def run(fused):
        NF = copy.copy(base)
        NF._model = copy.deepcopy(base._model)
        NF.best_model = NF._model
        NF.fused_minibatch = fused
        it = iter(draws)
        NF.sample_hook = lambda n, dd, dtype, stage: next(it)[:n].to(dtype)
        optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        res = NF._train_variance_forward_seq(f, optim, False, None, mbs * n_mb, 5, 0, True, True, mini_batch_size=mbs, integrate=True,
                                             preburn_time=2, loss_mode=loss_mode)
        return res, t2n(torch.stack(NF.history)), [t2n(p.grad) for p in NF._model.parameters()]

    ra, ha, ga = run(True)
    rb, hb, gb = run(False)
    np.testing.assert_allclose(ha, hb, rtol=1e-8)
    np.testing.assert_allclose(ra, rb, rtol=1e-5)
    gscale = max(float(np.abs(y).max()) for y in gb)         # (a Linear bias in front of a train-mode BatchNorm has gradient 0: what both
    for x, y in zip(ga, gb):                                  # paths return there is rounding noise relative to the OTHER gradients)
        np.testing.assert_allclose(x, y, rtol=1e-6, atol=1e-9 * max(1e-30, gscale) + 1e-14)


@pytest.mark.parametrize("case", range(8))
def test_random_rambo_with_pdf_grid_vs_oracle(case):
    """Seeded sweep of 2 -> n maps WITH the device-side PDF factor (n = 2 .. 10, massless / massive legs, cuts on / off, several parton pairs,
    ragged N): weights and momenta against the oracle, plane input bit-identical to row input (oracle/pdf.py: parity unpinned)."""
    from nf_amd import integrands
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    subs, grid = _synthetic_grid(True)
    rng = np.random.default_rng(900 + case)
    n = int(rng.integers(2, 11))
    final = [float(rng.uniform(5, 90)) if rng.random() < 0.5 else 0.0 for _ in range(n)]
    cuts = (-1, -1, -1) if rng.random() < 0.4 else (float(rng.uniform(5, 25)), float(rng.uniform(0.2, 0.5)), float(rng.uniform(2.5, 5)))
    pdgs = [[21, 21], [2, -1], [1, 21], [22, 21], [-2, 1]][int(rng.integers(0, 5))]
    E, N = float(rng.uniform(6000, 13000)), int(rng.integers(100, 6000))
    r = rng.random((N, 3 * n - 2))
    ref_m, ref_w = orambo.generate_kinematics([0., 0.], final, E, r, *cuts, pdf_active=True, tau=True, pdf=subs, pdgs=pdgs)
    gen = FlatInvertiblePhasespace([0., 0.], final, pdf=grid, pdf_active=True, tau=True)
    mom, wgt = gen.generateKinematics_batch(E, torch.as_tensor(r).to(dev()), *cuts, pdgs=pdgs)
    w = t2n(wgt)
    differ = (w == 0) != (ref_w == 0)
    assert differ.sum() <= 2
    np.testing.assert_allclose(w[~differ], ref_w[~differ], rtol=1e-8)
    np.testing.assert_allclose(t2n(mom), ref_m, rtol=1e-8, atol=1e-9 * E)
    planes = torch.as_tensor(r.T.copy()).to(dev())
    f = integrands.RamboWeight(gen, E, *cuts, pdgs=pdgs)
    np.testing.assert_array_equal(t2n(f.from_planes(planes, list(range(3 * n - 2)), N)), w)


@pytest.mark.parametrize("kind,d,P,K", [("pwquad", 3, 1, 5), ("pwlin", 4, 2, 9), ("pwquad", 6, 3, 8)])
@pytest.mark.parametrize("train", [True, False])
def test_conditioner_without_hidden_layers(kind, d, P, K, train):
    """NN_layers = []: the reference's layer list then holds ONE hidden layer as wide as the head (sizes = [T * out]: BatchNorm -> Linear ->
    BatchNorm -> ReLU -> Linear, coupling_cells.py:236-254) -- a matrix-core conditioner for all but the smallest cells; forward against
    the oracle, adjoint against autograd."""
    from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
    torch.manual_seed(d + K)
    cell = (PWQuad if kind == "pwquad" else PWLin)(d, P, K, []).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.add_(0.3 * torch.randn_like(p))
    params = {"c." + k: t2n(v) for k, v in cell.state_dict().items()}
    N = 777
    x = torch.rand(N, d + 1, dtype=torch.double) * (1 - 1e-12)
    x[:, -1] += 0.5
    want, upd = {}, {}
    y_ref = oflow.cell_forward(x.numpy(), params, "c", kind, P, K, train, upd, want)
    cell = cell.to(dev()).train(train)
    res = cell._own_engine().forward(x.to(dev()), True, train, want_xj=True, want_bins=True)
    np.testing.assert_array_equal(t2n(res["bins"]).T, want["bins"])
    np.testing.assert_allclose(t2n(res["xj"]), y_ref, rtol=RTOL, atol=ATOL)
    cell.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in params.items()})
    model = torch.nn.Sequential(cell).train(train)
    R = torch.randn(N, d + 1, dtype=torch.double, device=dev())

    def grads(force_torch):
        for p in cell.parameters():
            p.grad = None
        xx = x.to(dev()).clone().requires_grad_(True)
        out = model_torch(model, xx) if force_torch else cell(xx)
        (out * R).sum().backward()
        return xx.grad.clone(), [p.grad.clone() for p in cell.parameters()]

    sd = {k: v.clone() for k, v in cell.state_dict().items()}
    gx_h, gp_h = grads(False)
    cell.load_state_dict(sd)
    gx_t, gp_t = grads(True)
    np.testing.assert_allclose(t2n(gx_h), t2n(gx_t), rtol=1e-6, atol=1e-9 * max(1.0, float(gx_t.abs().max())))
    for (name, _), a, b in zip(cell.named_parameters(), gp_h, gp_t):
        np.testing.assert_allclose(t2n(a), t2n(b), rtol=1e-6, atol=1e-9 * max(1.0, float(b.abs().max())), err_msg=name)


@pytest.mark.parametrize("cfg,N,G", _random_models(8, 555))
def test_random_models_integrate_builtin_equals_callable(cfg, N, G):
    """`integrate()` end to end on the seeded model sweep: the built-in integrand (fused rider, or kept planes + plane integrand where the
    model has no rider) against the same function as a Python callable (kept planes + torch expression), identical draws."""
    from nf_amd import integrands
    from nf_amd.normalizing_flows.manager import AffineManager
    d = cfg["d"]
    torch.manual_seed(N)
    if cfg["kind"] == "affine":
        NF = AffineManager(n_flow=d)
        NF.create_model(cfg["P"], cfg["n_cells"], list(cfg["NN"]), cfg["roll"])
    else:
        NF = build_manager(cfg)
    sd0 = {k: v.clone() for k, v in NF._model.state_dict().items()}
    nitn = G + 2
    f = integrands.Normal(0.5, 0.35, d)

    def f_py(x):
        return torch.exp(-((x - 0.5) ** 2).sum(-1) / (2 * 0.35 ** 2)) / (0.35 * np.sqrt((2 * np.pi) ** d))

    gen = torch.Generator().manual_seed(N)
    draws = [(torch.rand(N, d, generator=gen) * (1 - 2 ** -23)).to(torch.float32).numpy() for _ in range(nitn)]
    NF.batch_iterations = G
    NF.sample_hook = Replay(draws)
    a = NF.integrate(f, nitn, N, 0)
    NF._model.load_state_dict(sd0)
    NF.sample_hook = Replay(draws)
    b = NF.integrate(f_py, nitn, N, 0)
    np.testing.assert_allclose(float(a[0]), float(b[0]), rtol=2e-6)
    np.testing.assert_allclose(float(a[1]), float(b[1]), rtol=2e-5)
