"""Managers with the reference's API (/root/reference/nisrep/normalizing_flows/manager.py):
`PWLinManager` / `PWQuadManager` `.create_model`, `BasicManager._train_variance_forward_seq`,
`BasicManager.integrate`, `.format_input`, `.model`, helpers `get_bin` and `normal`.

Host control flow (epochs, preburn, early stopping, best-model bookkeeping, logging, checkpoints)
stays Python exactly as in the reference; every tensor-sized computation goes to libnis_hip.so:
model forward (`FlowSequential`), `f(X)*J` for built-in integrands, mean / variance / max reductions
(`nis_moments`).  There is no CPU path: a ROCm GPU is required.

Deviations from the reference, on purpose:
  * PWLinManager casts its model to float64 (the reference forgets to, Q2, and then crashes).
  * no `gc.collect()` per minibatch (manager.py:270) and no tqdm dependency; `pretty_progressbar`
    is accepted and ignored when tqdm is missing.
  * random draws happen on the GPU generator unless `sample_hook` is set (tests replay the
    reference's CPU draws through it).
"""
import copy
import datetime
import math
import os

import numpy as np
import torch

from .. import _lib
from ..flow import moments as hip_moments, moments_of
from .layers.coupling_cells import AffineCoupling, PWLin, PWQuad
from .layers.layers import AddJacobian, DeMaskLayer, MaskLayer, RollLayer
from .sequential import FlowSequential
from ._autograd import flow_loss, flow_loss_batched


def get_bin(x, n=0):
    """Binary digits of x, most significant first, zero-filled to n digits (manager.py:20-36)."""
    return [int(c) for c in format(x, "b").zfill(n)]


def normal(x, mu, sigma, n_flow):
    """Gaussian test integrand (manager.py:39-40; sigma enters linearly in the norm, as there)."""
    return (torch.exp(-torch.sum((x - mu) ** 2 / (2 * sigma ** 2), -1))) / (sigma * np.sqrt((2 * np.pi) ** n_flow))


class _HipVar(torch.autograd.Function):
    """Unbiased variance (torch.var) with the reduction in nis_moments; d var / d v_i = 2 (v_i - mean)/(n-1)."""

    @staticmethod
    def forward(ctx, v):
        m = hip_moments(v)
        ctx.save_for_backward(v, m)
        return m[2] / (m[0] - 1.0)

    @staticmethod
    def backward(ctx, g):
        v, m = ctx.saved_tensors
        return (v - m[1]) * (2.0 * g / (m[0] - 1.0))


class _HipMeanSq(torch.autograd.Function):
    """mean(v**2) from the same moment record: M2/n + mean^2."""

    @staticmethod
    def forward(ctx, v):
        m = hip_moments(v)
        ctx.save_for_backward(v, m)
        return m[2] / m[0] + m[1] * m[1]

    @staticmethod
    def backward(ctx, g):
        v, m = ctx.saved_tensors
        return v * (2.0 * g / m[0])


def hip_var(v):
    return _HipVar.apply(v)


def hip_mean(v):
    return hip_moments(v)[1]


def hip_max(v):
    return hip_moments(v)[3]


class ModelAPI:
    @property
    def model(self):
        if self._model is not None:
            return self._model
        raise AttributeError("No model was instantiated")


class BasicManager(ModelAPI):
    """Training (variance loss, forward mode) and Monte-Carlo integration (manager.py:52-405)."""

    format_input = AddJacobian()

    def __init__(self, n_flow=2, *args):
        self.n_flow = n_flow
        self._model = None
        self._inverse_model = None
        self.optimizer_object = None
        self.best_model = None
        self.sample_hook = None      # callable(n, d, dtype, stage) -> tensor: replaces the uniform draws
        self.data_parallel = None    # nf_amd.parallel.DataParallel: gradient / bookkeeping all-reduce per optimizer step

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _device(dev):
        _lib.require_gpu()
        if isinstance(dev, torch.device):
            return dev
        return torch.device("cuda:" + str(0 if dev is None else dev))

    def _uniform(self, n, dev, dtype, stage):
        if self.sample_hook is not None:
            return self.sample_hook(n, self.n_flow, dtype, stage).to(dev)
        return torch.rand(n, self.n_flow, dtype=dtype, device=dev)

    def _stock_input(self):
        """True when `format_input` is the stock AddJacobian with a unit Jacobian: the HIP forward then implies that
        column instead of receiving an (N, d+1) tensor whose last column was copied from the host (layers.py:75-77)."""
        fi = self.format_input
        return type(fi) is AddJacobian and fi.jacobian_value.numel() == 1 and float(fi.jacobian_value) == 1.0

    def _forward(self, w, dev):
        if self._stock_input() and isinstance(self.model, FlowSequential):
            return self.model.forward_raw(w.to(dev))
        return self.model(self.format_input(w, dev))

    # ------------------------------------------------------------------ training
    def _train_variance_forward_seq(self, f, optimizer_object, log=True, logdir=None, batch_size=10000, epochs=10,
                                    epoch_start=0, pretty_progressbar=True, save_best=True, run=None, dev=0,
                                    mini_batch_size=2000, integrate=False, preburn_time=75, kill_counter=7,
                                    impr_ratio=1e-2, loss_mode="var"):
        """Variance-loss training with the Jacobian computed in the forward pass (manager.py:66-378)."""
        dev = self._device(dev)
        check_time = preburn_time if preburn_time > 10 else 50                      # :78-81
        if mini_batch_size > batch_size:
            mini_batch_size = batch_size
        n_minibatches = int(batch_size / mini_batch_size)
        batch_size = batch_size - (batch_size % mini_batch_size)
        filename = self._open_logs(log, logdir, run)

        integ = torch.zeros((epochs + 1,), device=dev)                              # float32, as the reference (:114-115)
        err = torch.zeros((epochs + 1,), device=dev)
        self.model.to(dev)
        self.best_loss = 0
        self.best_var = 0
        maxf = 0
        d = self.n_flow

        # ---- flat-sampling survey: d batches of 2*mbs points (:139-165)
        survey = []
        for _ in range(d):
            w = self._uniform(2 * mini_batch_size, dev, torch.double, "survey")
            fres = f(w)
            if self.data_parallel is not None:
                survey.append(fres.detach())
            m = hip_moments(fres)
            integ[0] += m[1] * (2 * mini_batch_size) / (d * 2 * mini_batch_size)
            err[0] += (m[2] / (m[0] - 1.0)) / d
            if m[3] > maxf:
                maxf = m[3].clone()
            if loss_mode == "var":
                self.best_loss = self.best_loss + hip_var((fres / maxf).detach()) / d
            elif loss_mode == "est":
                self.best_loss = self.best_loss + _HipMeanSq.apply(fres.detach()) / d
            else:
                print("Unknown loss function")
                return
            m2 = hip_moments(((fres / maxf) ** 2).detach())
            self.best_var += float((m2[2] / (m2[0] - 1.0)) / 2 * mini_batch_size)   # precedence as in :165

        if self.data_parallel is not None:                  # same normalisation / starting point on every rank
            dp = self.data_parallel
            maxf = dp.sync_max(torch.as_tensor(maxf, device=dev))
            # the survey's loss / variance are ratios to max f: with the rank-local running maximum the ranks would average
            # differently normalised numbers -- redo them with the global one before averaging
            self.best_loss, self.best_var = 0, 0
            for fres in survey:
                if loss_mode == "var":
                    self.best_loss = self.best_loss + hip_var(fres / maxf) / d
                else:
                    self.best_loss = self.best_loss + _HipMeanSq.apply(fres) / d
                m2 = hip_moments((fres / maxf) ** 2)
                self.best_var += float((m2[2] / (m2[0] - 1.0)) / 2 * mini_batch_size)
            self.best_loss = dp.mean(self.best_loss)
            self.best_var = float(dp.mean(torch.as_tensor(self.best_var, device=dev, dtype=torch.double)))
            integ[0], err[0] = dp.mean(integ[0]), dp.mean(err[0])
        if save_best or log:                                                        # :170-196
            XJ = self._forward(w, dev)
            X, J = XJ[:, :-1], XJ[:, -1]
            self.varJ = _HipMeanSq.apply(J.detach())
            dkl = torch.nn.KLDivLoss(reduction="batchmean")
            self.DKL = dkl(torch.log(X.detach() + 1e-45), w.to(dev)).detach()
            if self.data_parallel is not None:              # the forward above advanced the running statistics rank-locally
                self.data_parallel.sync_buffers(self.model)
            self.best_model = copy.deepcopy(self.model)
            self.best_epoch = 0
            self.best_time = 0
            self.best_loss_rel = torch.ones_like(self.best_loss)
            self.best_func_count = 2 * batch_size * d
            self.history = []
            del XJ, X, J
        if run is not None and log:
            run.log_scalar("training.int_loss", self.best_loss.tolist(), 0)
        self.int_loss = self.best_loss

        stale_save = 1000
        preburner = preburn_time > 0
        counter = 0
        last_loss = 1000
        i = epoch_start - 1
        loss = var = None
        for i in range(epoch_start, epoch_start + epochs):                           # :212
            loss = 0
            var = 0
            optimizer_object.zero_grad()
            done = 0
            fuse_ok = self.fused_minibatch and loss_mode in ("var", "est") and torch.is_tensor(maxf) and self._stock_input() and \
                isinstance(self.model, FlowSequential)
            while fuse_ok and done < n_minibatches:
                # The epoch's minibatches are independent given the parameters (the loss is their mean, backward runs once, :219-275):
                # G of them travel as the y dimension of every launch -- forward with history, integrand, loss bookkeeping and (in
                # loss.backward()) the adjoint are three library calls per G minibatches instead of ~60 launches per minibatch.
                G = min(n_minibatches - done, self._minibatches_per_call(mini_batch_size))
                if self.sample_hook is not None:
                    w3 = torch.stack([self._uniform(mini_batch_size, dev, torch.double, "train") for _ in range(G)])
                else:
                    w3 = torch.rand(G, mini_batch_size, d, dtype=torch.double, device=dev)
                fused = flow_loss_batched(self.model, w3, f, maxf, preburner, loss_mode == "est")
                if fused is None:                   # outside the fused class: the per-minibatch path below takes these draws
                    pending = list(w3)
                    fuse_ok = False
                    break
                loss_sum, t = fused                 # t (G, 8): rows {mean(a), var(a), loss, var(u^2), n, mean(u), -, -}
                ts = t.sum(0)
                integ[i + 1] += ts[0] / n_minibatches
                err[i + 1] += ts[1] / n_minibatches
                loss = loss + loss_sum
                var = var + ts[3] / mini_batch_size                                 # variance of the mean (:268)
                done += G
            else:
                pending = []
            for _ in range(n_minibatches - done):                                   # :219-270
                w = pending.pop(0) if pending else self._uniform(mini_batch_size, dev, torch.double, "train")
                fused = None
                if self.fused_minibatch and loss_mode in ("var", "est") and torch.is_tensor(maxf) and self._stock_input() and \
                        isinstance(self.model, FlowSequential):
                    fused = flow_loss(self.model, w.to(dev), f, maxf, preburner, loss_mode == "est")
                if fused is not None:               # forward, integrand, loss bookkeeping and (later) the adjoint: all in libnis_hip
                    loss_mb, t = fused              # t = {mean(a), var(a), loss, var(u^2), n, mean(u)}
                    integ[i + 1] += t[0] / n_minibatches
                    err[i + 1] += t[1] / n_minibatches
                    loss = loss + loss_mb
                    var = var + t[3] / mini_batch_size                              # variance of the mean (:268)
                    continue
                XJ = self._forward(w, dev)
                J = XJ[:, -1]
                if preburner:                                                       # f on the latent point (:237-242)
                    fres = f(w)
                    fXJ = fres * J / maxf
                    m = hip_moments(fres)
                else:
                    fres = f(XJ[:, :-1].detach()) * J
                    fXJ = fres / maxf
                    m = hip_moments(fres)
                integ[i + 1] += m[1] / n_minibatches
                err[i + 1] += (m[2] / (m[0] - 1.0)) / n_minibatches
                if loss_mode == "var":
                    loss = loss + hip_var(fXJ)
                elif loss_mode == "est":
                    loss = loss + _HipMeanSq.apply(fXJ * maxf)
                m2 = hip_moments((fXJ ** 2).detach())
                var = var + (m2[2] / (m2[0] - 1.0)) / mini_batch_size               # variance of the mean (:268)
                del XJ, fXJ
            loss = loss / n_minibatches
            loss.backward()
            if self.data_parallel is not None:              # nf_amd.parallel.DataParallel: one all-reduce per step
                sc = self.data_parallel.sync_step(self.model, torch.stack(
                    [loss.detach().double(), torch.as_tensor(var, device=dev).double(), integ[i + 1].double(),
                     err[i + 1].double()]))
                loss, var = sc[0].to(loss.dtype), sc[1]
                integ[i + 1], err[i + 1] = sc[2].to(integ.dtype), sc[3].to(err.dtype)
            optimizer_object.step()
            self.history.append(loss.detach())
            if run is not None and log:
                run.log_scalar("training.loss", loss.tolist(), i)
                run.log_scalar("training.loss_rel", (loss / self.int_loss).tolist(), i)
            if save_best or log:
                self.best_func_count = self.best_func_count + batch_size
            if (save_best or log) and (loss < self.best_loss) and not preburner:    # :293-304
                self.best_loss = loss
                self.best_var = float(var)
                self.best_loss_rel = loss / self.int_loss
                if self.data_parallel is not None:          # every rank takes this branch together (loss is the synced one)
                    self.data_parallel.sync_buffers(self.model)
                self.best_model = copy.deepcopy(self.model)
                self.best_epoch = i
                self.best_time = (datetime.datetime.utcnow() - run.start_time).total_seconds() if run is not None else 0
            if loss < last_loss:                                                    # :307-316
                counter = 0
            else:
                counter += 1
                if counter > kill_counter and preburner:
                    counter = 0
                    preburner = False
                elif counter > kill_counter:
                    break
            last_loss = loss
            if (i % check_time == 0) and i > (preburn_time + 1) and float(self.best_loss / stale_save) > (1 - impr_ratio) \
                    and not preburner:
                break
            elif i % check_time == 0 and not preburner and (self.best_loss < self.int_loss or i > 300):
                stale_save = self.best_loss
            if preburner and ((loss < 0.25 * self.best_loss) or i > preburn_time):
                preburner = False

        endpoint = i + 1
        if self.data_parallel is not None:                  # leave the live model identical on every rank too
            self.data_parallel.sync_buffers(self.model)
        with torch.no_grad():                                                       # :333-346
            if integrate and endpoint < epochs - 1:
                model = self.best_model.eval()
                for s in range(endpoint, epochs):
                    for _ in range(n_minibatches):
                        w = self._uniform(mini_batch_size, dev, torch.double, "post")
                        XJ = model(self.format_input(w, dev)).detach()
                        m = hip_moments(f(XJ[:, :-1]), XJ[:, -1])
                        integ[s + 1] += m[1] / (n_minibatches * math.sqrt(mini_batch_size))
                        err[s + 1] += torch.sqrt(m[2] / (m[0] - 1.0)) / n_minibatches
                    self.best_func_count = self.best_func_count + batch_size

        self.integ_tot = torch.sum(integ / err) / torch.sum(1 / err)                 # :349-350
        self.err_tot = torch.sqrt(1 / torch.sum(1 / err))
        if run is not None and integrate:
            run.log_scalar("training.integ", self.integ_tot.tolist(), 0)
            run.log_scalar("training.err", self.err_tot.tolist(), 0)
        try:
            if log:
                torch.save({"best_epoch": self.best_epoch, "best_loss": self.best_loss, "int_loss": self.int_loss,
                            "best_loss_rel": self.best_loss_rel, "best_func_count": self.best_func_count,
                            "model_state_dict": self.best_model.state_dict(), "integ": self.integ_tot,
                            "err": self.err_tot}, filename)
        except Exception:
            print("Torch save not possible")
        if integrate:
            return (self.integ_tot.detach().tolist(), self.err_tot.detach().tolist())
        return (0, 0)

    def _open_logs(self, log, logdir, run):
        """Log-file names and the initial state_dict dump (manager.py:89-112)."""
        if logdir is None:
            return None
        if run is not None and log:
            base = logdir + "/" + str(run._id)
        else:
            base = logdir
        filename = base + "/torch"
        try:
            if not os.path.exists(base):
                os.makedirs(base)
            if log:
                open(filename, "w+").close()
                open(filename + "_int", "w+").close()
                torch.save({"model_state_dict": self.best_model.state_dict()}, filename + "_int")
        except Exception:
            print("Torch save not possible")
        return filename

    # ------------------------------------------------------------------ checkpoints
    def load_checkpoint(self, filename, map_location=None):
        """Load a checkpoint the REFERENCE (or this package) wrote into the instantiated model: `logdir/torch_int`
        ({'model_state_dict'}, manager.py:107-109) or `logdir/torch` (+ best_epoch, best_loss, int_loss, best_loss_rel,
        best_func_count, integ, err; manager.py:360-369).  The state_dict key layout is the reference's, so models trained
        there integrate here and vice versa.  Returns the loaded dict."""
        model = self.model                                   # AttributeError("No model was instantiated") like the reference
        ck = torch.load(filename, map_location=map_location or "cpu", weights_only=False)
        model.load_state_dict(ck["model_state_dict"])
        self.best_model = model
        for key in ("best_epoch", "best_loss", "int_loss", "best_loss_rel", "best_func_count"):
            if key in ck:
                setattr(self, key, ck[key])
        if "integ" in ck:
            self.integ_tot, self.err_tot = ck["integ"], ck["err"]
        return ck

    def save_training_state(self, filename, optimizer_object, epoch):
        """Everything needed to continue `_train_variance_forward_seq` later (the reference can only restart from
        scratch): model, best model, optimizer state, bookkeeping, and the epoch to continue from."""
        keep = {k: getattr(self, k) for k in ("best_epoch", "best_loss", "int_loss", "best_loss_rel", "best_func_count", "best_var",
                                              "history") if hasattr(self, k)}
        torch.save({"model_state_dict": self.model.state_dict(),
                    "best_model_state_dict": (self.best_model or self.model).state_dict(),
                    "optimizer_state_dict": optimizer_object.state_dict(), "epoch": int(epoch), **keep}, filename)

    def load_training_state(self, filename, optimizer_object, map_location=None):
        """Inverse of `save_training_state`; returns the epoch to pass as `epoch_start`."""
        ck = torch.load(filename, map_location=map_location or "cpu", weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])
        self.best_model = copy.deepcopy(self.model)
        self.best_model.load_state_dict(ck["best_model_state_dict"])
        optimizer_object.load_state_dict(ck["optimizer_state_dict"])
        for key in ("best_epoch", "best_loss", "int_loss", "best_loss_rel", "best_func_count", "best_var", "history"):
            if key in ck:
                setattr(self, key, ck[key])
        return int(ck["epoch"])

    # ------------------------------------------------------------------ integration
    def integrate(self, f, nitn, neval, dev=None):
        """Inverse-variance weighted Monte-Carlo estimate over nitn batches of neval points drawn in
        float32 (manager.py:380-405).  Returns two 0-d float32 CPU tensors (sig, sig_err).

        `f` may be a `nf_amd.integrands` object (fused HIP evaluation straight from the column planes)
        or any callable on a (neval, d) float64 CUDA tensor."""
        if self.best_model is None:
            print("No model has been trained")
            return (0, 0)
        dev = self._device(dev if dev is not None else self._model_device())
        neval, nitn = int(neval), int(nitn)
        eng = self.best_model.engine() if isinstance(self.best_model, FlowSequential) else None
        if eng is not None:
            eng.clear_bin_ambiguous()
        mom = self.moments_batches(f, nitn, neval, dev)                             # (nitn, 4) on the device
        mom = mom.cpu()                                                             # the only host sync
        # did any PWQuad bin decision of this call rest on last bits (include/nis_hip.h: NIS_STATUS_BIN_AMBIGUOUS)?  Everywhere else the
        # bins are the reference's, bit for bit.  Read behind the results' own synchronisation: a few bytes.
        self.bin_ambiguous = bool(eng.bin_ambiguous()) if eng is not None else False
        var = (mom[:, 2] / (mom[:, 0] - 1.0)).to(torch.float32)                     # float32 stores (:391-392)
        mean = mom[:, 1].to(torch.float32)
        sig = torch.sum(mean / var) / torch.sum(1 / var)
        sig_err = torch.sqrt(1 / torch.sum(1 / var)) / np.sqrt(neval * nitn)
        return (sig, sig_err)

    def _model_device(self):
        p = next(self.best_model.parameters())
        return p.device if p.is_cuda else torch.device("cuda:0")

    def _minibatches_per_call(self, mini_batch_size):
        """How many minibatches one fused training call carries: all of an epoch's, within ~2 GiB of history planes."""
        n_cells = sum(1 for m in self.model if hasattr(m, "KIND")) if isinstance(self.model, FlowSequential) else 1
        per = (n_cells + 1) * (self.n_flow + 1) * int(mini_batch_size) * 8
        return int(max(1, min(self.minibatches_per_call, (2 << 30) // max(per, 1))))

    minibatches_per_call = 256   # training: upper bound on the minibatches of one optimizer step carried by one fused call (the launch
                                 # cost is amortised long before; the record workspace grows with the count)
    batch_iterations = 20      # integrate(): iterations fused into one launch set (independent BN batches)
    batch_callable = True      # integrate() with a Python callable: the flow part of G iterations as one launch set (False: one
                               # forward per iteration with a materialised (N, d+1) XJ, the round-1 path kept for A/B tests)
    fused_minibatch = True     # training: forward -> integrand -> loss bookkeeping -> adjoint as library calls only (False: the piecewise
                               # path through torch's autograd graph, kept for A/B tests and for models outside the adjoint kernels' class)

    def moments_batches(self, f, nitn, neval, dev):
        """(count, mean, M2, max) of f(X)*J for nitn fresh batches; stays on the device."""
        model = self.best_model
        model.to(dev)
        out = torch.empty((nitn, 4), dtype=torch.double, device=dev)
        flow = isinstance(model, FlowSequential)
        fusable = flow and model.engine().fusable()
        # the draw happens inside the step (manager.py:395) when nobody replays given draws: the first launch of a set generates the
        # float32 batch itself (Philox, keyed from torch's generator so torch.manual_seed governs it); no torch.rand launch, no input
        in_kernel = self.sample_hook is None and (neval >= 2 or not model.training)
        if fusable and hasattr(f, "nis_kind"):      # G iterations per library call, X / J never stored
            eng = model.engine()
            G = max(1, min(int(self.batch_iterations), nitn, max(1, (1 << 28) // max(1, neval * (self.n_flow + 1)))))
            plan = None
            if in_kernel:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                # whole sets of the short-kernel d = 2 chain are replayed from a hipGraph (flow.IntegratePlan)
                if nitn >= G and model.training and os.environ.get("NIS_NO_GRAPH", "0") in ("", "0") and \
                        eng.rng_mode(G, neval, True, f.nis_kind) == 2:
                    plan = eng.plan(G, neval, dev, f.nis_kind, f.nis_params)
                    plan.reset(seed, 0)
            i = 0
            while i < nitn:
                g = min(G, nitn - i)
                if plan is not None and g == G:
                    plan.run(out[i:i + g])
                elif in_kernel:
                    eng.integrate_steps_rng(g, neval, dev, f.nis_kind, f.nis_params, out[i:i + g], seed, i, train=model.training)
                else:
                    w = torch.stack([self._uniform(neval, dev, torch.float32, "integrate") for _ in range(g)])
                    eng.integrate_steps(w, model.training, f.nis_kind, f.nis_params, out[i:i + g])
                i += g
            return out
        planar = hasattr(f, "from_planes") and flow
        if planar and fusable and hasattr(f, "moments_from_planes"):
            # e.g. RAMBO weight: reads the flow's column planes, X is never assembled, no weights are written.  Per launch set of G
            # iterations: the flow keeps its planes (the rider integrates f == 1, i.e. <J>, and is ignored), then ONE launch turns
            # the G plane sets into weight * J moment records.  The status word accumulates on the device and is read once.
            eng = model.engine()
            G = max(1, min(int(self.batch_iterations), nitn, max(1, (1 << 27) // max(1, neval * (self.n_flow + 1)))))
            # kept planes live for this call only (torch's caching allocator makes the re-allocation free; a buffer cached on the
            # manager held 0.5 - 0.8 GB away from the training workspaces for the manager's lifetime: ADVICE r3)
            scratch = torch.empty((G, 4), dtype=torch.double, device=dev)
            planes = torch.empty((G, eng.d + 1, neval), dtype=torch.double, device=dev)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            if in_kernel:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            i = 0
            while i < nitn:
                g = min(G, nitn - i)
                if in_kernel:
                    eng.integrate_steps_rng(g, neval, dev, _lib.NIS_F_ONE, [], scratch[:g], seed, i, train=model.training,
                                            planes=planes[:g], keep_planes=True)
                else:
                    w = torch.stack([self._uniform(neval, dev, torch.float32, "integrate") for _ in range(g)])
                    eng.integrate_steps(w, model.training, _lib.NIS_F_ONE, [], scratch[:g], planes=planes[:g])
                f.moments_from_planes(planes[:g], eng.out_col, eng.d, neval, out[i:i + g], status)
                i += g
            if not f.check_status(status):                 # the ONE host read of the call
                return out
            # some sample had x1 == x2: the reference skips the lab boost for that WHOLE batch -- redo batch by batch below
        if flow and not planar and self._stock_input() and self.batch_callable:
            # Any Python callable (the reference's README form, manager.py:397-400): the FLOW part of G iterations is still one launch
            # set with the draw inside (NIS_F_NONE: X and J stay in the column planes, nothing else is written); f then sees X of
            # one iteration at a time, as in the reference's loop -- a (neval, d) float64 view of the planes (like the reference's
            # XJ[:, :-1] it is not contiguous), and one reduction launch pair turns f(X) * J into the moment record.  Also the way of a
            # built-in integrand on a model whose last cell has no fused rider (matrix-core PWQuad with a bin count other than
            # 4 / 8 / 16 / 32): the integrand is then evaluated from the kept planes.
            eng = model.engine()
            d = eng.d
            G = max(1, min(int(self.batch_iterations), nitn, max(1, (1 << 27) // max(1, neval * (d + 1)))))
            planes = torch.empty((G, d + 1, neval), dtype=torch.double, device=dev)        # this call's only (see above)
            identity = list(eng.out_col) == list(range(d))
            cols = None if identity else torch.as_tensor(eng.out_col, device=dev)
            if in_kernel:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            i = 0
            with torch.no_grad():
                while i < nitn:
                    g = min(G, nitn - i)
                    if in_kernel:
                        eng.integrate_steps_rng(g, neval, dev, _lib.NIS_F_NONE, [], None, seed, i, train=model.training,
                                                planes=planes[:g], keep_planes=True)
                    else:
                        w = torch.stack([self._uniform(neval, dev, torch.float32, "integrate") for _ in range(g)])
                        eng.integrate_steps(w, model.training, _lib.NIS_F_NONE, [], None, planes=planes[:g])
                    for gi in range(g):
                        if hasattr(f, "nis_kind"):
                            out[i + gi] = moments_of(planes[gi], d, eng.out_col, f.nis_kind, f.nis_params)[0]
                            continue
                        X = planes[gi, :d].t() if identity else planes[gi].index_select(0, cols).t()
                        out[i + gi] = hip_moments(f(X), planes[gi, d])
                    i += g
            return out
        for i in range(nitn):
            w = self._uniform(neval, dev, torch.float32, "integrate")
            if planar:                      # e.g. RAMBO weight: reads the flow's column planes, X is never assembled
                res = model.flow_forward(w)
                eng = model.engine()
                out[i] = hip_moments(f.from_planes(res["planes"], eng.out_col, neval), res["planes"][eng.d])
                continue
            with torch.no_grad():
                if self._stock_input() and flow:
                    XJ = model.flow_forward(w, want_xj=True)["xj"]
                else:
                    XJ = model(self.format_input(w, dev)).detach()
                out[i] = hip_moments(f(XJ[:, :-1]), XJ[:, -1])
        return out


class AffineManager(BasicManager):
    """Affine coupling cells with cyclic rolls (manager.py:411-453).  As in PWLinManager every roll is registered under
    the name "roll", so one RollLayer (after cell 0) survives.  The reference never casts this model to float64 while
    AddJacobian promotes the input, so its own warm-up forward raises "mixed dtype"; like PWLinManager's counterpart
     this build casts to float64 -- recorded deviation."""

    format_input = AddJacobian()

    def create_model(self, n_pass_through, n_cells, NN, roll_step):
        self._model = FlowSequential()
        for i_cell in range(n_cells):
            self._model.add_module(str(i_cell), AffineCoupling(flow_size=self.n_flow, pass_through_size=n_pass_through,
                                                               NN_layers=NN))
            self._model.add_module("roll", RollLayer(roll_step))
        self._finish_model(0)


class PWLinManager(BasicManager):
    """Piecewise-linear cells; one cyclic roll after cell 0 (all rolls share the child name "roll" in the
    reference, so only one survives: manager.py:487-492)."""

    format_input = AddJacobian()

    def create_model(self, n_pass_through, n_cells, n_bins, NN, roll_step):
        self._model = FlowSequential()
        for i_cell in range(n_cells):
            self._model.add_module(str(i_cell), PWLin(flow_size=self.n_flow, pass_through_size=n_pass_through,
                                                      n_bins=n_bins, NN_layers=NN))
            self._model.add_module("roll", RollLayer(roll_step))
        self._finish_model(0)


class PWQuadManager(BasicManager):
    """Piecewise-quadratic cells with roll (d <= 7) or binary-mask (d > 7) permutations (manager.py:518-600)."""

    format_input = AddJacobian()

    def create_model(self, n_cells, n_bins, NN, dev=0):
        d = self.n_flow
        if n_cells < 2 * np.ceil(np.log2(d)) and n_cells < d:                        # :526-534
            n_cells = d if d <= 6 else 6 if d == 7 else int(2 * np.ceil(np.log2(d)))
            print("Adjusted # coupling cells to " + str(n_cells))
        self._model = FlowSequential()
        if d <= 7:
            n_pass_through = 1 if d <= 6 else 2
            for i in range(n_cells):
                self._model.add_module(str(i), PWQuad(flow_size=d, pass_through_size=n_pass_through, n_bins=n_bins,
                                                      NN_layers=NN))
                shift = 1 if i < n_cells - 1 else d - ((n_cells - 1) % d)            # total roll = multiple of d
                self._model.add_module("roll" + str(i), RollLayer(shift))
        else:
            n = len(get_bin(d - 1, 0))
            dims_bin = torch.IntTensor([get_bin(i, n) for i in range(d)])
            for i in range(2 * n):
                masker = MaskLayer(dims_bin, i)
                self._model.add_module("mask" + str(i), masker)
                self._model.add_module(str(i), PWQuad(flow_size=d, pass_through_size=masker.pass_through,
                                                      n_bins=n_bins, NN_layers=NN))
                self._model.add_module("demask" + str(i), DeMaskLayer(masker.feeder, masker.trafoer))
            rest = n_cells - 2 * n
            for i in range(rest):
                self._model.add_module(str(i + 2 * n), PWQuad(flow_size=d, pass_through_size=int(d / 2), n_bins=n_bins,
                                                              NN_layers=NN))
                shift = 1 if i < rest - 1 else d - ((rest - 1) % d)
                self._model.add_module("roll" + str(i + 2 * n), RollLayer(shift))
        self._finish_model(dev)


def _finish_model(self, dev):
    """Common tail of create_model (manager.py:493-499, :589-598): float64 on the device, alias
    best_model, and one warm-up forward on 5 points in train mode (it updates the BN running stats)."""
    dev = self._device(dev)
    self._model.to(dev).to(torch.double)
    self.best_model = self.model
    w = self._uniform(5, dev, torch.double, "warmup")
    with torch.no_grad():
        self._model(self.format_input(w, dev))


BasicManager._finish_model = _finish_model
