#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING the
reference package (NGoetz/NF, `nisrep`) on the CPU of the build container.

    python tests/golden/gen_golden.py            # needs /root/reference

The reference cannot travel to the GPU box, so only DATA produced here is
committed: inputs, parameters (state_dict arrays) and the outputs the
reference computed for them.  Nothing in this file restates the reference's
algorithm; intermediate values (conditioner output, bin indices, bisection
roots, random draws) are captured by hooking torch functions / module hooks
while the unmodified reference code runs.

Output files (np.savez_compressed, keys are "<case>/<name>"):
    cells.npz      single coupling cells, PWQuad / PWLin, train- and eval-mode BN
    models.npz     whole PWQuadManager / PWLinManager models (topology + XJ)
    integrate.npz  BasicManager.integrate runs (captured w draws, per-iteration mean/var)
    training.npz   short _train_variance_forward_seq runs (history, grads, returns)
    rambo.npz      FlatInvertiblePhasespace.generateKinematics_batch cases
    affine.npz     AffineCoupling cells and AffineManager models
    wide.npz       C3 / C4 / C5 topologies (width-32 conditioners): autograd gradients at N = 512, train-mode forward at N = 2e5
"""
import os
import sys
import types
import json
import tempfile
import contextlib

import numpy as np
import torch

REF = os.environ.get("NIS_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
# `lhapdf` is imported by the reference constructor when pdf_active=True even if
# pdf=None (no lhapdf function is ever called in that case) -> empty stub module.
sys.modules.setdefault("lhapdf", types.ModuleType("lhapdf"))

from nisrep.normalizing_flows.manager import PWQuadManager, PWLinManager, AffineManager, normal  # noqa: E402
from nisrep.normalizing_flows.layers.coupling_cells import PWQuad, PWLin, AffineCoupling  # noqa: E402
from nisrep.normalizing_flows.layers.layers import AddJacobian  # noqa: E402
from nisrep.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def npy(t):
    return t.detach().cpu().numpy().copy()


def put_state(out, prefix, module):
    for k, v in module.state_dict().items():
        out[f"{prefix}/sd/{k}"] = npy(v)


@contextlib.contextmanager
def capture(fn_owner, fn_name, store):
    """Record every return value of fn_owner.fn_name while the block runs."""
    orig = getattr(fn_owner, fn_name)

    def wrapped(*a, **k):
        r = orig(*a, **k)
        store.append(r.detach().clone() if torch.is_tensor(r) else r)
        return r

    setattr(fn_owner, fn_name, wrapped)
    try:
        yield
    finally:
        setattr(fn_owner, fn_name, orig)


def randomize(module, gen_seed, last_scale=2.0):
    """Move the parameters / BN buffers away from their init values so that the
    vectors are sensitive to every term (gamma, beta, running stats, bias)."""
    g = torch.Generator().manual_seed(gen_seed)
    mods = list(module.NN)
    lin = [m for m in mods if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for m in mods:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(1.0 + 0.3 * (torch.rand(m.weight.shape, generator=g, dtype=torch.double) - 0.5))
                m.bias.copy_(0.4 * (torch.rand(m.bias.shape, generator=g, dtype=torch.double) - 0.5))
                m.running_mean.copy_(0.5 * torch.rand(m.running_mean.shape, generator=g, dtype=torch.double))
                m.running_var.copy_(0.05 + torch.rand(m.running_var.shape, generator=g, dtype=torch.double))
        lin[-1].weight.mul_(last_scale)
        lin[-1].bias.copy_(torch.rand(lin[-1].bias.shape, generator=g, dtype=torch.double) - 0.5)


# --------------------------------------------------------------------------- cells
def gen_cells():
    out = {}
    meta = {}
    shapes = [(2, 1, 4, [3] * 3), (5, 2, 6, [7, 7]), (8, 4, 32, [32] * 3), (16, 8, 16, [32] * 3), (3, 1, 5, [4])]
    for kind, cls in (("pwquad", PWQuad), ("pwlin", PWLin)):
        for si, (d, P, K, NN) in enumerate(shapes):
            for N in (7, 64, 1024):
                if N == 1024 and si not in (0, 2):
                    continue
                for mode in ("train", "eval"):
                    name = f"{kind}_d{d}_P{P}_K{K}_N{N}_{mode}"
                    torch.manual_seed(1000 + 17 * si + N)
                    cell = cls(d, P, K, list(NN)).double()
                    randomize(cell, 77 + si)
                    x = torch.rand(N, d + 1, dtype=torch.double)
                    x[:, -1] = 0.5 + torch.rand(N, dtype=torch.double)  # running Jacobian column
                    if kind == "pwquad":
                        # edge values of the transformed block: 0, 1-1e-6, above the clamp, exactly 1
                        x[0, P] = 0.0
                        x[1, P] = 1.0 - 1e-6
                        x[2, P] = 1.0 - 1e-9
                        x[3, d - 1] = 1.0
                    else:
                        x[0, P] = 0.0
                        x[1, P] = 1.0 - 1e-12   # PWLin: x == 1.0 is an index error in the reference
                    put_state(out, name, cell)
                    out[f"{name}/x"] = npy(x)
                    cell.train(mode == "train")
                    zs, idx = [], []
                    h = cell.NN.register_forward_hook(lambda m, i, o: zs.append(o.detach().clone()))
                    hook_fn = "argmax" if kind == "pwquad" else "floor"
                    with capture(torch, hook_fn, idx):
                        y = cell(x.clone())
                    h.remove()
                    out[f"{name}/y"] = npy(y)
                    out[f"{name}/bins"] = npy(idx[0]).astype(np.int64)
                    if N <= 64:
                        out[f"{name}/Z"] = npy(zs[0])
                    for k, v in cell.state_dict().items():
                        if "running" in k or "num_batches" in k:
                            out[f"{name}/sd_after/{k}"] = npy(v)
                    meta[name] = dict(kind=kind, d=d, P=P, K=K, NN=list(NN), N=N, mode=mode)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "cells.npz"), **out)
    print("cells.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- whole models
def build_pwquad(d, n_cells, K, NN, seed):
    torch.manual_seed(seed)
    NF = PWQuadManager(n_flow=d)
    NF.create_model(n_cells, K, list(NN))
    return NF


def build_pwlin(d, P, n_cells, K, NN, roll, seed):
    torch.manual_seed(seed)
    NF = PWLinManager(n_flow=d)
    try:
        NF.create_model(P, n_cells, K, list(NN), roll)
    except RuntimeError:
        # reference quirk: the model is never cast to double, so its own warm-up
        # forward raises "mixed dtype"; _model/best_model are already assigned.
        pass
    NF._model.double()
    return NF


def topo(model):
    names, passes = [], []
    for n, m in model.named_children():
        names.append(n + ":" + type(m).__name__)
        if hasattr(m, "pass_through_size"):
            passes.append(int(m.pass_through_size))
    return names, passes


def gen_models():
    out, meta = {}, {}
    cases = [("pwquad", dict(d=2, n_cells=2, K=4, NN=[3] * 3)),
             ("pwquad", dict(d=5, n_cells=2, K=6, NN=[8, 8])),
             ("pwquad", dict(d=7, n_cells=3, K=5, NN=[6])),
             ("pwquad", dict(d=8, n_cells=2, K=8, NN=[16] * 2)),
             ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32] * 3)),
             ("pwquad", dict(d=16, n_cells=8, K=16, NN=[32] * 3)),
             ("pwquad", dict(d=9, n_cells=10, K=4, NN=[5])),
             ("pwlin", dict(d=8, P=4, n_cells=4, K=64, NN=[32] * 3, roll=4)),
             ("pwlin", dict(d=3, P=1, n_cells=3, K=7, NN=[5, 5], roll=1))]
    for ci, (kind, c) in enumerate(cases):
        name = f"{kind}_model_d{c['d']}_c{c['n_cells']}"
        if kind == "pwquad":
            NF = build_pwquad(c["d"], c["n_cells"], c["K"], c["NN"], 50 + ci)
        else:
            NF = build_pwlin(c["d"], c["P"], c["n_cells"], c["K"], c["NN"], c["roll"], 50 + ci)
        names, passes = topo(NF._model)
        put_state(out, name, NF._model)
        N = 257
        w = torch.rand(N, c["d"], dtype=torch.double)
        out[f"{name}/w"] = npy(w)
        NF._model.train()
        out[f"{name}/XJ_train"] = npy(NF._model(NF.format_input(w, torch.device("cpu"))))
        for k, v in NF._model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{name}/sd_after/{k}"] = npy(v)
        NF._model.eval()
        out[f"{name}/XJ_eval"] = npy(NF._model(NF.format_input(w, torch.device("cpu"))))
        meta[name] = dict(kind=kind, children=names, passes=passes, N=N, **c)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    print("models.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- integrate
def camel(x):
    return torch.exp(-((x[:, 0] - 0.75) ** 2 + (x[:, 1] - 0.75) ** 2) / (0.2 ** 2)) + \
        torch.exp(-((x[:, 0] - 0.25) ** 2 + (x[:, 1] - 0.25) ** 2) / (0.2 ** 2))


def gen_integrate():
    out, meta = {}, {}
    runs = [("camel_d2", "pwquad", dict(d=2, n_cells=2, K=4, NN=[3] * 3), camel, 3, 2000),
            ("normal_d8", "pwlin", dict(d=8, P=4, n_cells=4, K=16, NN=[16] * 2, roll=4),
             lambda x: normal(x, 0.5, 0.25, 8), 2, 1500)]
    for name, kind, c, f, nitn, neval in runs:
        if kind == "pwquad":
            NF = build_pwquad(c["d"], c["n_cells"], c["K"], c["NN"], 7)
        else:
            NF = build_pwlin(c["d"], c["P"], c["n_cells"], c["K"], c["NN"], c["roll"], 7)
        put_state(out, name, NF._model)
        draws, xjs = [], []
        h = NF.best_model.register_forward_hook(lambda m, i, o: xjs.append(o.detach().clone()))
        torch.manual_seed(123)
        with capture(torch.nn.init, "uniform_", draws):
            sig, sig_err = NF.integrate(f, nitn, neval)
        h.remove()
        means, variances = [], []
        for XJ in xjs:
            fres = f(XJ[:, :-1]) * XJ[:, -1]
            means.append(float(torch.mean(fres)))
            variances.append(float(torch.var(fres)))
        out[f"{name}/w"] = np.stack([npy(d) for d in draws])           # (nitn, neval, d) float32
        out[f"{name}/XJ_last"] = npy(xjs[-1])
        out[f"{name}/mean"] = np.array(means)
        out[f"{name}/var"] = np.array(variances)
        out[f"{name}/sig"] = npy(sig)
        out[f"{name}/sig_err"] = npy(sig_err)
        for k, v in NF.best_model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{name}/sd_after/{k}"] = npy(v)
        meta[name] = dict(kind=kind, nitn=nitn, neval=neval, seed=123, **c)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "integrate.npz"), **out)
    print("integrate.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- training
class RecordingAdamax(torch.optim.Adamax):
    """torch.optim.Adamax that keeps a copy of every gradient it is asked to apply."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.seen = []

    def step(self, closure=None):
        self.seen.append([None if p.grad is None else p.grad.detach().clone()
                          for g in self.param_groups for p in g["params"]])
        return super().step(closure)


def gen_training():
    out, meta = {}, {}
    runs = [("var_mbs512", dict(batch=512, mbs=512, loss_mode="var", epochs=4, preburn=1)),
            ("var_mbs128", dict(batch=512, mbs=128, loss_mode="var", epochs=4, preburn=1)),
            ("est_mbs256", dict(batch=512, mbs=256, loss_mode="est", epochs=3, preburn=0))]
    for name, c in runs:
        NF = build_pwquad(2, 2, 4, [3] * 3, 11)
        put_state(out, name + "/init", NF._model)
        opt = RecordingAdamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
        draws = []
        torch.manual_seed(321)
        with tempfile.TemporaryDirectory() as td, capture(torch.nn.init, "uniform_", draws):
            ret = NF._train_variance_forward_seq(camel, opt, True, td, c["batch"], c["epochs"], 0, False, True,
                                                 mini_batch_size=c["mbs"], integrate=True,
                                                 preburn_time=c["preburn"], loss_mode=c["loss_mode"])
            ck = torch.load(os.path.join(td, "torch"), weights_only=False)
            meta_keys = sorted(ck.keys())
        out[f"{name}/history"] = np.array([float(h) for h in NF.history])
        out[f"{name}/ret"] = np.array(ret, dtype=np.float64)
        for att in ("best_loss", "best_var", "varJ", "DKL", "int_loss", "best_loss_rel", "integ_tot", "err_tot"):
            out[f"{name}/{att}"] = np.array(float(getattr(NF, att)))
        out[f"{name}/best_epoch"] = np.array(int(NF.best_epoch))
        out[f"{name}/best_func_count"] = np.array(float(NF.best_func_count))
        for di, dr in enumerate(draws):
            out[f"{name}/draw{di:03d}"] = npy(dr)
        pnames = [k for k, _ in NF._model.named_parameters()]
        for ei, grads in enumerate(opt.seen):
            for pn, g in zip(pnames, grads):
                out[f"{name}/grad{ei}/{pn}"] = npy(g)
        put_state(out, name + "/final", NF._model)
        put_state(out, name + "/best", NF.best_model)
        meta[name] = dict(ckpt_keys=meta_keys, n_draws=len(draws), pnames=pnames, seed=321, model_seed=11, **c)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "training.npz"), **out)
    print("training.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- RAMBO
def gen_rambo():
    out, meta = {}, {}
    cases = [
        ("m2to4_massive", dict(initial=[100., 100.], final=[100.] * 4, pdf_active=False, tau=True,
                               E_cm=1000., cuts=(-1, -1, -1), N=193)),
        ("m2to4_massive_cuts0", dict(initial=[100., 100.], final=[100.] * 4, pdf_active=False, tau=True,
                                     E_cm=1000., cuts=(0, 0, -1), N=61)),
        ("m2to4_tau", dict(initial=[0., 0.], final=[100.] * 4, pdf_active=True, tau=True,
                           E_cm=13000., cuts=(0, 0, -1), N=193)),
        ("m2to6_massless_cuts", dict(initial=[0., 0.], final=[0.] * 6, pdf_active=True, tau=True,
                                     E_cm=13000., cuts=(20., 0.4, 2.5), N=193)),
        ("m2to2_massless", dict(initial=[0., 0.], final=[0., 0.], pdf_active=False, tau=True,
                                E_cm=500., cuts=(-1, -1, -1), N=33)),
        ("m2to3_mixed", dict(initial=[0., 5.], final=[10., 0., 30.], pdf_active=False, tau=True,
                             E_cm=400., cuts=(5., 0.3, 3.0), N=65)),
        ("m2to3_x1x2", dict(initial=[0., 0.], final=[0., 50., 0.], pdf_active=True, tau=False,
                            E_cm=8000., cuts=(10., 0.4, 4.0), N=65)),
        ("m2to5_massive_tau", dict(initial=[0., 0.], final=[173., 173., 0., 80., 4.], pdf_active=True, tau=True,
                                   E_cm=13000., cuts=(15., 0.4, 5.0), N=129)),
    ]
    for si, (name, c) in enumerate(cases):
        gen = FlatInvertiblePhasespace(c["initial"], c["final"], pdf=None, pdf_active=c["pdf_active"], tau=c["tau"])
        nd = gen.nDimPhaseSpace() + (2 if c["pdf_active"] else 0)
        torch.manual_seed(900 + si)
        r = torch.rand(c["N"], nd, dtype=torch.double)
        us = []
        with capture(gen, "bisect_vec_batch", us):
            mom, wgt = gen.generateKinematics_batch(c["E_cm"], r.clone(), pT_mincut=c["cuts"][0],
                                                    delR_mincut=c["cuts"][1], rap_maxcut=c["cuts"][2], pdgs=[2, -1])
        out[f"{name}/r"] = npy(r)
        out[f"{name}/momenta"] = npy(mom)
        out[f"{name}/weight"] = npy(wgt)
        if us and us[0] is not None:
            out[f"{name}/u"] = npy(us[0])
        meta[name] = dict(c, ndim=nd, flat=float(FlatInvertiblePhasespace.get_flatWeights(c["E_cm"], len(c["final"]))))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "rambo.npz"), **out)
    print("rambo.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- affine cells / models
def gen_affine():
    out, meta = {}, {}
    shapes = [(2, 1, [3] * 3), (5, 2, [7, 7]), (8, 4, [32] * 3), (6, 5, [4]), (9, 3, [20, 12])]
    for si, (d, P, NN) in enumerate(shapes):
        for N in (7, 64, 1024):
            if N == 1024 and si not in (0, 2):
                continue
            for mode in ("train", "eval"):
                name = f"affine_d{d}_P{P}_N{N}_{mode}"
                torch.manual_seed(3000 + 13 * si + N)
                cell = AffineCoupling(d, P, list(NN)).double()
                randomize(cell, 177 + si, last_scale=1.0)
                x = torch.rand(N, d + 1, dtype=torch.double)
                x[:, -1] = 0.5 + torch.rand(N, dtype=torch.double)
                x[0, P] = 0.0
                x[1, P] = 1.0
                put_state(out, name, cell)
                out[f"{name}/x"] = npy(x)
                cell.train(mode == "train")
                zs = []
                h = cell.NN.register_forward_hook(lambda m, i, o: zs.append(o.detach().clone()))
                y = cell(x.clone())
                h.remove()
                out[f"{name}/y"] = npy(y)
                if N <= 64:
                    out[f"{name}/Z"] = npy(zs[0])
                for k, v in cell.state_dict().items():
                    if "running" in k or "num_batches" in k:
                        out[f"{name}/sd_after/{k}"] = npy(v)
                meta[name] = dict(kind="affine", what="cell", d=d, P=P, NN=list(NN), N=N, mode=mode)
    models = [dict(d=2, P=1, n_cells=2, NN=[3] * 3, roll=1), dict(d=5, P=2, n_cells=3, NN=[8, 8], roll=2),
              dict(d=8, P=4, n_cells=4, NN=[32] * 3, roll=4)]
    for ci, c in enumerate(models):
        name = f"affine_model_d{c['d']}_c{c['n_cells']}"
        torch.manual_seed(90 + ci)
        NF = AffineManager(n_flow=c["d"])
        try:
            NF.create_model(c["P"], c["n_cells"], list(c["NN"]), c["roll"])
        except RuntimeError:
            # same reference quirk as PWLinManager: the model is never cast to double while AddJacobian promotes the
            # input to float64, so the warm-up forward raises "mixed dtype"; _model is already assigned.
            pass
        NF._model.double()
        names, passes = topo(NF._model)
        put_state(out, name, NF._model)
        N = 257
        w = torch.rand(N, c["d"], dtype=torch.double)
        out[f"{name}/w"] = npy(w)
        NF._model.train()
        out[f"{name}/XJ_train"] = npy(NF._model(NF.format_input(w, torch.device("cpu"))))
        for k, v in NF._model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{name}/sd_after/{k}"] = npy(v)
        NF._model.eval()
        out[f"{name}/XJ_eval"] = npy(NF._model(NF.format_input(w, torch.device("cpu"))))
        meta[name] = dict(kind="affine", what="model", children=names, passes=passes, N=N, **c)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "affine.npz"), **out)
    print("affine.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- wide conditioners (BASELINE configs C3 / C4 / C5)
def kronecker_w(N, d):
    """Deterministic, exactly reproducible inputs in [0, 1): element k = i d + j of the golden-ratio Kronecker sequence,
    ((k + 1) * 0x9E3779B97F4A7C15 mod 2^64) >> 11, times 2^-53 (the tests rebuild it with the same integer arithmetic)."""
    k = np.arange(1, N * d + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = (k * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(11)
    return (v.astype(np.float64) * 2.0 ** -53).reshape(N, d)


def gen_wide():
    """(a) reference autograd gradients of every parameter (and of the input) for a fixed random cotangent on XJ, C3 and C5
    topologies at N = 512, train-mode BN;  (b) train-mode forward of the C3 / C4 / C5 topologies at N = 200 000: per-column
    moments of XJ, 4 096 sampled rows and the updated running statistics (the input is kronecker_w, not stored)."""
    out, meta = {}, {}
    topo_cases = {"C3": ("pwlin", dict(d=8, P=4, n_cells=4, K=64, NN=[32] * 3, roll=4)),
                  "C4": ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32] * 3)),
                  "C5": ("pwquad", dict(d=16, n_cells=8, K=16, NN=[32] * 3))}

    def build(kind, c, seed):
        if kind == "pwquad":
            return build_pwquad(c["d"], c["n_cells"], c["K"], c["NN"], seed)
        return build_pwlin(c["d"], c["P"], c["n_cells"], c["K"], c["NN"], c["roll"], seed)

    for tag in ("C3", "C5"):
        kind, c = topo_cases[tag]
        name = f"grad_{tag}"
        NF = build(kind, c, 70)
        model = NF._model
        for i, m in enumerate(model.children()):
            if hasattr(m, "NN"):
                randomize(m, 700 + i, last_scale=1.0)
        put_state(out, name, model)
        N = 512
        g = torch.Generator().manual_seed(71)
        w = torch.rand(N, c["d"], dtype=torch.double, generator=g).requires_grad_(True)
        cot = torch.randn(N, c["d"] + 1, dtype=torch.double, generator=g)
        model.train()
        XJ = model(NF.format_input(w, torch.device("cpu")))
        (XJ * cot).sum().backward()
        out[f"{name}/w"] = npy(w)
        out[f"{name}/cot"] = npy(cot)
        out[f"{name}/XJ"] = npy(XJ)
        out[f"{name}/gw"] = npy(w.grad)
        for pn, p in model.named_parameters():
            out[f"{name}/grad/{pn}"] = npy(p.grad)
        names, passes = topo(model)
        meta[name] = dict(kind=kind, what="grad", N=N, children=names, passes=passes, **c)

    torch.set_num_threads(8)
    for tag in ("C3", "C4", "C5"):
        kind, c = topo_cases[tag]
        name = f"fwd_{tag}"
        NF = build(kind, c, 80)
        model = NF._model
        for i, m in enumerate(model.children()):
            if hasattr(m, "NN"):
                randomize(m, 800 + i, last_scale=1.0)
        put_state(out, name, model)
        N = 200_000
        w = torch.as_tensor(kronecker_w(N, c["d"]))
        model.train()
        with torch.no_grad():
            XJ = model(NF.format_input(w, torch.device("cpu")))
        rows = np.arange(0, N, N // 4096)[:4096]
        out[f"{name}/rows"] = rows
        out[f"{name}/XJ_rows"] = npy(XJ[rows])
        out[f"{name}/col_mean"] = npy(XJ.mean(0))
        out[f"{name}/col_var"] = npy(XJ.var(0))
        out[f"{name}/col_min"] = npy(XJ.min(0).values)
        out[f"{name}/col_max"] = npy(XJ.max(0).values)
        for k, v in model.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"{name}/sd_after/{k}"] = npy(v)
        names, passes = topo(model)
        meta[name] = dict(kind=kind, what="fwd", N=N, children=names, passes=passes, **c)
        print(name, "done", flush=True)
    torch.set_num_threads(1)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "wide.npz"), **out)
    print("wide.npz:", len(meta), "cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["cells", "models", "integrate", "training", "rambo", "affine", "wide"]
    for w in which:
        globals()["gen_" + w]()
