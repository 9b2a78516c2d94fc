"""Host engine: turns a Sequential of coupling cells + permutation layers into one
`nis_flow` handle and drives libnis_hip.so (no arithmetic happens in Python / torch here).

Reference call sites this replaces: `self.model(self.format_input(w, dev))`
(/root/reference/nisrep/normalizing_flows/manager.py:174-178, :225-229, :341, :396-397).
"""
import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import CellDesc


def _param_tensors(cell):
    """Parameters / BN buffers of a cell in C-ABI blob order (= order of the reference's nn.Sequential)."""
    # the nn.Parameter OBJECTS of a module survive .to() / .double() / load_state_dict (their .data is swapped), so the walk over
    # the conditioner is done once per set of parameter objects (the training step asks four times per optimizer step)
    cache = getattr(cell, "_pt_cache", None)
    if cache is not None and all(a is b for a, b in zip(cache[3], cell.NN._modules.values())) and len(cache[3]) == len(cell.NN._modules) \
            and all(p is q for p, q in zip(cache[0], cache[4]())):
        return cache[0], cache[1], cache[2]
    bns = [m for m in cell.NN if isinstance(m, torch.nn.BatchNorm1d)]
    lins = [m for m in cell.NN if isinstance(m, torch.nn.Linear)]
    ps, rs = [], []
    getters = []
    for l, bn in enumerate(bns):
        ps += [bn.weight, bn.bias, lins[l].weight]
        getters += [(bn, "weight"), (bn, "bias"), (lins[l], "weight")]
        if l < len(bns) - 1 and lins[l].bias is not None:       # AffineCoupling: biased hidden Linears
            ps.append(lins[l].bias)
            getters.append((lins[l], "bias"))
        rs += [(bn, "running_mean"), (bn, "running_var")]
    ps.append(lins[-1].bias)
    getters.append((lins[-1], "bias"))

    def current(getters=getters):
        return [m._parameters[n] for m, n in getters]

    cell._pt_cache = (ps, rs, bns, list(cell.NN._modules.values()), current)
    return ps, rs, bns


def _same_device(a, b):
    a, b = torch.device(a), torch.device(b)
    return a.type == b.type and (a.index or 0) == (b.index or 0)


def _is_flat(cell, dev):
    flat, run = getattr(cell, "_flat", None), getattr(cell, "_running", None)
    if flat is None or run is None or not _same_device(flat.device, dev):
        return False
    ps, rs, _ = _param_tensors(cell)
    base, off = flat.data_ptr(), 0
    for p in ps:
        if p.data_ptr() != base + 8 * off or p.dtype != torch.double:
            return False
        off += p.numel()
    base, off = run.data_ptr(), 0
    for bn, name in rs:
        t = getattr(bn, name)
        if t.data_ptr() != base + 8 * off or t.dtype != torch.double:
            return False
        off += t.numel()
    return True


def ensure_flat(cell, device=None):
    """Keep one contiguous float64 blob per cell for parameters and one for BN running statistics
    (layout: include/nis_hip.h).  The nn.Parameters / buffers are VIEWS into the blobs, so optimisers,
    state_dict(), load_state_dict() keep working while the kernels read a single pointer.  .to(),
    .double() and deepcopy break the aliasing; it is detected here (pointer check) and rebuilt."""
    ps, rs, bns = _param_tensors(cell)
    dev = torch.device(device) if device is not None else ps[0].device
    if _is_flat(cell, dev):
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).to(device=dev, dtype=torch.double) for p in ps])
        run = torch.cat([getattr(bn, n).detach().reshape(-1).to(device=dev, dtype=torch.double) for bn, n in rs])
        trk = torch.stack([bn.num_batches_tracked.detach().to(dev).to(torch.long) for bn in bns])
        off = 0
        for p in ps:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            if p.grad is not None:
                p.grad = p.grad.to(device=dev, dtype=torch.double)
            off += n
        off = 0
        for bn, name in rs:
            n = getattr(bn, name).numel()
            bn._buffers[name] = run[off:off + n]
            off += n
        for i, bn in enumerate(bns):
            bn._buffers["num_batches_tracked"] = trk[i]
    cell._flat, cell._running, cell._tracked = flat, run, trk


def cell_desc(cell, in_col):
    d = CellDesc()
    d.kind = cell.KIND
    d.d = cell.flow_size
    d.P = cell.pass_through_size
    d.K = int(getattr(cell, "n_bins", 0) or 0)
    widths = cell.hidden_widths
    d.n_hidden = len(widths)
    if len(widths) > _lib.NIS_MAX_HIDDEN:
        raise _lib.NisError(f"conditioner depth {len(widths)} > {_lib.NIS_MAX_HIDDEN}")
    for i, w in enumerate(widths):
        d.width[i] = int(w)
    for j, p in enumerate(in_col):
        d.in_col[j] = int(p)
    return d


class FlowEngine:
    """One `nis_flow` handle for an ordered list of (cell, in_col) + the final column map.  Owns the
    persistent device workspace (zero-padded parameter image, BN coefficients, partial records): its size
    does not depend on the batch, and keeping it lets libnis_hip skip re-packing unchanged parameters."""

    def __init__(self, cells, in_cols, out_col, d):
        self.L = _lib.lib()
        self.cells = list(cells)
        self.d = d
        self.out_col = list(out_col)
        n = len(self.cells)
        arr = (CellDesc * n)(*[cell_desc(c, ic) for c, ic in zip(self.cells, in_cols)])
        oc = (C.c_int32 * d)(*self.out_col)
        h = C.c_void_p()
        _lib.check(self.L.nis_flow_create(arr, n, d, oc, C.byref(h)), "nis_flow_create")
        self.handle = h
        self._descs = arr
        self.n_bins_rows = sum(c.transform_size for c in self.cells)
        self._finalizer = weakref.finalize(self, self.L.nis_flow_destroy, h)
        self._ws = None
        self._sig = None
        self._token = 0

    def fusable(self):
        """True when a built-in integrand rides in the last cell's kernel: the narrow (register width <= 8) fp64 VALU apply
        kernels, and the FUSE instantiations of the matrix-core apply kernel for wide conditioners."""
        return bool(self.L.nis_flow_fusable(self.handle))

    # ---- per-call plumbing
    def _pointers(self, dev):
        n = len(self.cells)
        pp, rp, tp = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        for i, c in enumerate(self.cells):
            ensure_flat(c, dev)
            pp[i], rp[i], tp[i] = c._flat.data_ptr(), c._running.data_ptr(), c._tracked.data_ptr()
        return pp, rp, tp

    def _workspace(self, dev, G=1):
        wsb = self.L.nis_flow_workspace_bytes(self.handle, G)
        if self._ws is None or self._ws.device != dev or self._ws.numel() < wsb:
            self._ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            self._ws[:8].zero_()                     # status words: [0] per call, [1] sticky (include/nis_hip.h)
            self._sig = None
        return self._ws, wsb

    def _status_buffers(self):
        bufs = [self._ws] if self._ws is not None else []
        plans = getattr(self, "_plans", None)
        if plans is not None:
            bufs.append(plans[1].ws)
        return bufs

    def clear_bin_ambiguous(self):
        """Clear the STICKY status word of every workspace this engine launches into (the kernels only ever OR into it)."""
        for b in self._status_buffers():
            b[4:8].zero_()

    def bin_ambiguous(self):
        """True when some PWQuad bin decision since the last `clear_bin_ambiguous()` rested on last bits (NIS_STATUS_BIN_AMBIGUOUS:
        an edge quotient within 2^-50 of x, or a width on the threshold of vanishing at the decision point -- the samples on which the
        reference's own CPU and GPU builds disagree).  Synchronises (one 4-byte read per workspace)."""
        return any(int(b[4:8].view(torch.int32)[0].item()) & _lib.NIS_STATUS_BIN_AMBIGUOUS for b in self._status_buffers())

    # The running statistics live in blobs the kernels update behind torch's back; the count of such updates is kept ON THE
    # CELL (several engines can share a cell: `model[0](x)` through the cell's own engine and `model(x)` through the
    # Sequential's), so an engine sees the updates another engine made and re-packs its eval-mode BatchNorm image.
    @property
    def _stats_epoch(self):
        return sum(getattr(c, "_stats_epoch", 0) for c in self.cells)

    def _stats_bump(self, d):
        for c in self.cells:
            c._stats_epoch = getattr(c, "_stats_epoch", 0) + d

    def pack_token(self):
        """Changes whenever a parameter / BN buffer was modified (torch version counters) or a train-mode
        forward (of ANY engine over these cells) advanced the running statistics: libnis_hip then re-packs its parameter image."""
        # (the tensors behind a cell's state_dict are looked up once per blob generation: building eight state_dicts per call cost
        #  more host time than the launches of a small integrate() set)
        tens = []
        for c in self.cells:
            cache = getattr(c, "_sd_cache", None)
            if cache is None or cache[0] is not c._flat or cache[1] is not c._running:
                cache = (c._flat, c._running, list(c.state_dict(keep_vars=True).values()))
                c._sd_cache = cache
            tens.append(cache[2])
        sig = tuple(getattr(c, "_stats_epoch", 0) for c in self.cells) + tuple(t._version for ts in tens for t in ts)
        if sig != self._sig:
            self._sig = sig
            self._token += 1
        return self._token

    def _check_input(self, w, cols, train):
        _lib.require_gpu()
        if not w.is_cuda:
            raise _lib.NisError("nf_amd kernels need the sample batch on the GPU (no CPU fallback)")
        if w.dtype not in (torch.float32, torch.float64):
            w = w.to(torch.double)
        w = w.contiguous()
        if w.dim() != 2 or w.shape[1] != cols:
            raise ValueError(f"expected input of shape (N,{cols}), got {tuple(w.shape)}")
        if train and w.shape[0] < 2:
            raise ValueError("Expected more than 1 value per channel when training")   # torch BatchNorm1d error
        return w

    def forward(self, w, has_jac, train, want_xj=True, want_bins=False, token=None):
        """w: (N,d) or (N,d+1) CUDA tensor (float32/float64).  Returns dict(planes, xj, bins, status)."""
        w = self._check_input(w, self.d + (1 if has_jac else 0), train)
        N, dev = w.shape[0], w.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev)
        if token is None:
            token = self.pack_token()
        planes = torch.empty((self.d + 1, N), dtype=torch.double, device=dev)
        xj = torch.empty((N, self.d + 1), dtype=torch.double, device=dev) if want_xj else None
        bins = torch.empty((self.n_bins_rows, N), dtype=torch.int32, device=dev) if want_bins else None
        with _lib.on_device(dev):
            rc = self.L.nis_flow_forward(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                         1 if has_jac else 0, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL,
                                         token, planes.data_ptr(), N, xj.data_ptr() if want_xj else None,
                                         bins.data_ptr() if want_bins else None, ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_forward")
        if train:
            self._stats_bump(1)
        status = ws[:4].view(torch.int32)
        self._raise_on_status(status)
        ambiguous = None
        if want_bins:                                # entries are bin | NIS_BIN_AMBIGUOUS_FLAG (include/nis_hip.h)
            ambiguous = (bins & _lib.NIS_BIN_AMBIGUOUS_FLAG) != 0
            bins = bins & ~_lib.NIS_BIN_AMBIGUOUS_FLAG
        return dict(planes=planes, xj=xj, bins=bins, ambiguous=ambiguous, status=status, ws=ws)

    def forward_sync(self, w, has_jac=False, want_xj=True, group=None):
        """Train-mode forward of THIS RANK'S rows of a batch that is sharded over the ranks of `group`, with the BatchNorm
        statistics of the whole batch (SURVEY.md 8(e) option 1): `nis_flow_forward_sync`, whose exchange hook is one
        in-place all-gather (RCCL on GPUs) of `world` x (3 x width) doubles per BatchNorm layer.  Returns the same dict as
        `forward`; concatenating the ranks' results reproduces the single-process forward of the concatenated batch."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # every check that can fail happens BEFORE the first exchange, and the ranks agree on the outcome with one small
        # all-reduce: a rank that raised alone would leave its peers blocked in the first all-gather
        err = None
        try:
            w = self._check_input(w, self.d + (1 if has_jac else 0), False)      # a shard may hold a single row
            if max([c.pass_through_size for c in self.cells] + [max(c.hidden_widths) for c in self.cells]) > 64:
                raise _lib.NisError("forward_sync exchanges at most 64 features per BatchNorm layer")
        except Exception as exc:                          # noqa: BLE001 - re-raised below on every rank
            err = exc
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=w.device if torch.is_tensor(w) and w.is_cuda else "cpu")
        if dist.get_backend(group) == "nccl" and not ok.is_cuda:
            ok = ok.cuda()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if err is not None:
            raise err
        if int(ok.item()) == 0:
            raise _lib.NisError("forward_sync: another rank rejected its arguments; no exchange was started")
        N, dev = w.shape[0], w.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev)
        planes = torch.empty((self.d + 1, N), dtype=torch.double, device=dev)
        xj = torch.empty((N, self.d + 1), dtype=torch.double, device=dev) if want_xj else None
        buf = torch.zeros(world * 64 * 3, dtype=torch.double, device=dev)

        failure = []

        def exchange(ctx, records, count, stream):
            try:                                             # an exception must not vanish inside the ctypes callback
                assert count <= 64 * 3, f"record of {count} doubles exceeds the exchange buffer"
                flat = buf[:world * count]
                dist.all_gather_into_tensor(flat, flat[rank * count:(rank + 1) * count].clone(), group=group)
            except BaseException as exc:                     # noqa: BLE001 - re-raised after the library call returns
                failure.append(exc)

        hook = _lib.SYNC_FN(exchange)
        with _lib.on_device(dev):
            rc = self.L.nis_flow_forward_sync(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                              1 if has_jac else 0, N, pp, rp, tp, self.pack_token(), planes.data_ptr(), N,
                                              xj.data_ptr() if want_xj else None, ws.data_ptr(), wsb, _lib.stream_ptr(dev), hook, None,
                                              buf.data_ptr(), world, rank)
        if failure:
            raise failure[0]
        _lib.check(rc, "nis_flow_forward_sync")
        self._stats_bump(1)
        status = ws[:4].view(torch.int32)
        self._raise_on_status(status)
        return dict(planes=planes, xj=xj, bins=None, status=status, ws=ws)

    def _raise_on_status(self, status):
        """PWLin with x == 1.0 indexes bin n_bins: the reference's torch.gather raises (coupling_cells.py:136); the kernel
        flags it instead.  Only models with a PWLin cell pay the device->host read of the status word."""
        if any(c.KIND == _lib.NIS_PWLIN for c in self.cells) and int(status[0].item()) & _lib.NIS_STATUS_PWLIN_RANGE:
            raise IndexError("index out of range in PWLin bin lookup (an input coordinate equals 1.0 or is negative)")

    # ---- training: forward with history + hand-written adjoint
    def backward_supported(self):
        """True when every cell is in the class the adjoint kernels cover (include/nis_hip.h)."""
        return bool(self.L.nis_flow_backward_supported(self.handle))

    def forward_train(self, w, has_jac, train=True):
        """Forward that keeps what the adjoint needs (train-mode BatchNorm, or eval mode: running statistics).  Returns (xj, saved)."""
        w = self._check_input(w, self.d + (1 if has_jac else 0), train)
        N, dev = w.shape[0], w.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev)
        hist = torch.empty((len(self.cells) + 1, self.d + 1, N), dtype=torch.double, device=dev)
        bnsave = torch.empty(self.L.nis_flow_bnsave_count(self.handle), dtype=torch.double, device=dev)
        xj = torch.empty((N, self.d + 1), dtype=torch.double, device=dev)
        with _lib.on_device(dev):
            rc = self.L.nis_flow_forward_hist(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                              1 if has_jac else 0, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL,
                                              self.pack_token(), hist.data_ptr(), N, bnsave.data_ptr(), xj.data_ptr(), ws.data_ptr(), wsb,
                                              _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_forward_hist")
        if train:
            self._stats_bump(1)
        self._raise_on_status(ws[:4].view(torch.int32))
        return xj, (hist, bnsave, has_jac, bool(train))

    def _bwd_workspace(self, N, dev):
        """Grow-only scratch of the adjoint (gradient planes, per-cell buffers), kept between calls: one minibatch's backward
        has finished with it before the next one starts on the same stream."""
        wsb = self.L.nis_flow_backward_workspace_bytes(self.handle, N)
        ws = getattr(self, "_bws", None)
        if ws is None or ws.device != dev or ws.numel() < wsb:
            self._bws = ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        return ws, wsb

    def loss_terms(self, xj, maxf, preburn, est, f=None, fvals=None, w=None):
        """The minibatch's loss bookkeeping in two launches (`nis_loss_terms`, manager.py:223-268).  xj: (N, d+1) model output;
        maxf: 0-d float64 device tensor; f: built-in integrand evaluated in the kernel on X (or on the latent rows `w` during the
        preburn), or fvals: (N,) values of an arbitrary callable.  Returns (fvals, rec12, out8) device tensors."""
        N, dev = xj.shape[0], xj.device
        d = xj.shape[1] - 1
        rec = torch.empty(12, dtype=torch.double, device=dev)
        out = torch.empty(8, dtype=torch.double, device=dev)
        ws = _moments_ws(dev, self.L.nis_loss_workspace_bytes())
        if fvals is None:
            rows = w if preburn else xj
            fv = torch.empty(N, dtype=torch.double, device=dev)
            fp = (C.c_double * 4)(*(list(f.nis_params) + [0.0] * 4)[:4])
            args = (None, rows.data_ptr(), rows.stride(0), d, f.nis_kind, fp, fv.data_ptr())
        else:
            fv = fvals.detach().to(torch.double).contiguous()
            args = (fv.data_ptr(), None, 0, d, -1, None, None)
        with _lib.on_device(dev):
            rc = self.L.nis_loss_terms(*args, xj.data_ptr() + 8 * d, d + 1, N, maxf.data_ptr(), 1 if preburn else 0, 1 if est else 0,
                                       rec.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "nis_loss_terms")
        return fv, rec, out

    def backward_loss(self, saved, fvals, maxf, rec, gloss, est):
        """Adjoint seeded by the minibatch loss (`nis_flow_backward_loss`): [flat parameter-gradient blob per cell]."""
        hist, bnsave, has_jac, _train = saved
        N, dev = hist.shape[2], hist.device
        pp, _, _ = self._pointers(dev)
        n = len(self.cells)
        gflat = [torch.empty_like(c._flat) for c in self.cells]
        gp = (C.c_void_p * n)(*[g.data_ptr() for g in gflat])
        ws, wsb = self._bwd_workspace(N, dev)
        gloss = gloss.detach().to(device=dev, dtype=torch.double).reshape(1)
        with _lib.on_device(dev):
            rc = self.L.nis_flow_backward_loss(self.handle, hist.data_ptr(), N, N, 1 if has_jac else 0, pp, bnsave.data_ptr(),
                                               fvals.data_ptr(), maxf.data_ptr(), rec.data_ptr(), gloss.data_ptr(), 1 if est else 0, gp,
                                               ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_backward_loss")
        return gflat

    # ---- the G minibatches of one optimizer step per library call (manager.py:219-270)
    def forward_train_batched(self, w, train=True, want_xj=False):
        """w: (G, N, d) float64 / float32 CUDA tensor, one minibatch per g (own BatchNorm batch).  Returns (xj (G, N, d+1) or None,
        saved) with saved = (hist (G, n_cells+1, d+1, N), bnsave, train)."""
        if w.dim() != 3:
            raise ValueError(f"expected minibatches of shape (G, N, {self.d}), got {tuple(w.shape)}")
        G, N = int(w.shape[0]), int(w.shape[1])
        w2 = self._check_input(w.reshape(G * N, w.shape[2]), self.d, False)
        if train and N < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        dev = w2.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev, G)
        hist = torch.empty((G, len(self.cells) + 1, self.d + 1, N), dtype=torch.double, device=dev)
        nbn = self.L.nis_flow_bnsave_count(self.handle)
        bnsave = torch.empty((G if train else 1, nbn), dtype=torch.double, device=dev)
        xj = torch.empty((G, N, self.d + 1), dtype=torch.double, device=dev) if want_xj else None
        with _lib.on_device(dev):
            rc = self.L.nis_flow_forward_hist_batched(self.handle, w2.data_ptr(), _lib.NIS_F32 if w2.dtype == torch.float32 else _lib.NIS_F64,
                                                      G, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL, self.pack_token(),
                                                      hist.data_ptr(), N, bnsave.data_ptr(), xj.data_ptr() if want_xj else None,
                                                      ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_forward_hist_batched")
        if train:
            self._stats_bump(1)
        self._raise_on_status(ws[:4].view(torch.int32))
        return xj, (hist, bnsave, bool(train))

    def loss_terms_batched(self, saved, maxf, preburn, est, f=None, fvals=None, w=None):
        """The loss bookkeeping of G minibatches in two launches (`nis_loss_terms_batched`, manager.py:223-268).  A built-in
        integrand `f` is evaluated in the kernel straight from the output planes of the history (or on the latent rows `w`
        (G, N, d) float64 during the preburn); fvals: (G, N) values of an arbitrary callable.  Returns (fv (G, N), rec (G, 12), out (G, 8))."""
        hist = saved[0]
        G, nset, dd, N = hist.shape
        d, dev = dd - 1, hist.device
        set_ = dd * N
        rec = torch.empty((G, 12), dtype=torch.double, device=dev)
        out = torch.empty((G, 8), dtype=torch.double, device=dev)
        wsb = self.L.nis_loss_workspace_bytes_batched(G, N)
        ws = _moments_ws(dev, wsb)
        if fvals is None:
            fv = torch.empty((G, N), dtype=torch.double, device=dev)
            fp = (C.c_double * 4)(*(list(f.nis_params) + [0.0] * 4)[:4])
            if preburn:
                if w.dtype != torch.double or not w.is_contiguous():
                    w = w.to(torch.double).contiguous()
                src = (w.data_ptr(), 1, d, N * d, None)
                keep = w
            else:
                oc = (C.c_int32 * d)(*self.out_col)
                src = (hist.data_ptr() + 8 * (nset - 1) * set_, N, 1, nset * set_, oc)
                keep = None
            args = (None,) + src + (d, f.nis_kind, fp, fv.data_ptr())
        else:
            fv = fvals.detach().to(torch.double).contiguous()
            keep = None
            args = (fv.data_ptr(), None, 0, 0, 0, None, d, -1, None, None)
        with _lib.on_device(dev):
            rc = self.L.nis_loss_terms_batched(*args, hist.data_ptr() + 8 * ((nset - 1) * set_ + d * N), 1, nset * set_, G, N,
                                               maxf.data_ptr(), 1 if preburn else 0, 1 if est else 0, rec.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
        del keep
        _lib.check(rc, "nis_loss_terms_batched")
        return fv, rec, out

    def backward_loss_batched(self, saved, fvals, maxf, rec, gloss, est):
        """Adjoint of G minibatches seeded by their losses (`nis_flow_backward_loss_batched`): [flat parameter-gradient blob per cell],
        each the sum over the minibatches of gloss * d loss_g / d parameters."""
        hist, bnsave, train = saved
        G, _, _, N = hist.shape
        dev = hist.device
        pp, _, _ = self._pointers(dev)
        n = len(self.cells)
        gflat = [torch.empty_like(c._flat) for c in self.cells]
        gp = (C.c_void_p * n)(*[g.data_ptr() for g in gflat])
        wsb = self.L.nis_flow_backward_workspace_bytes_batched(self.handle, G, N)
        ws = getattr(self, "_bws", None)
        if ws is None or ws.device != dev or ws.numel() < wsb:
            self._bws = ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        gloss = gloss.detach().to(device=dev, dtype=torch.double).reshape(1)
        with _lib.on_device(dev):
            rc = self.L.nis_flow_backward_loss_batched(self.handle, hist.data_ptr(), N, G, N, pp, bnsave.data_ptr(),
                                                       _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL, fvals.data_ptr(), maxf.data_ptr(),
                                                       rec.data_ptr(), gloss.data_ptr(), 1 if est else 0, gp, ws.data_ptr(), ws.numel(),
                                                       _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_backward_loss_batched")
        return gflat

    def backward(self, saved, gxj, want_gx):
        """dL/dXJ (N, d+1) -> ([flat parameter-gradient blob per cell], dL/dx or None)."""
        hist, bnsave, has_jac, train = saved
        N, dev = hist.shape[2], hist.device
        pp, _, _ = self._pointers(dev)
        n = len(self.cells)
        gflat = [torch.empty_like(c._flat) for c in self.cells]
        gp = (C.c_void_p * n)(*[g.data_ptr() for g in gflat])
        ws, wsb = self._bwd_workspace(N, dev)
        gx = torch.empty((N, self.d + 1), dtype=torch.double, device=dev) if want_gx else None
        gxj = gxj.to(device=dev, dtype=torch.double).contiguous()
        with _lib.on_device(dev):
            fn = self.L.nis_flow_backward if train else self.L.nis_flow_backward_eval
            rc = fn(self.handle, hist.data_ptr(), N, N, 1 if has_jac else 0, pp, bnsave.data_ptr(), gxj.data_ptr(),
                    gp, gx.data_ptr() if want_gx else None, ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_flow_backward")
        return gflat, gx

    def integrate_steps(self, w, train, f_kind, f_params, out4, planes=None, token=None, want_fres=False):
        """G iterations of integrate()'s loop body in ONE library call.  w: (G, N, d) (or (N, d) for G = 1),
        out4: (G, 4) float64 device tensor/view receiving {count, mean, M2, max} per iteration.  The iterations
        run as the y dimension of every launch (each is its own train-mode BatchNorm batch); X and J are not
        stored unless `planes` ((G, d+1, N)) is passed."""
        if w.dim() == 2:
            w = w.unsqueeze(0)
        G, N = w.shape[0], w.shape[1]
        w2 = self._check_input(w.reshape(G * N, w.shape[2]), self.d, train)
        if train and N < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        dev = w2.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev, G)
        if token is None:
            token = self.pack_token()
        keep = planes is not None
        if planes is None:
            planes = self._planes_scratch(G, N, dev)
        fres = torch.empty((G, N), dtype=torch.double, device=dev) if want_fres else None
        fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
        with _lib.on_device(dev):
            rc = self.L.nis_integrate_steps(self.handle, w2.data_ptr(), _lib.NIS_F32 if w2.dtype == torch.float32 else _lib.NIS_F64,
                                            G, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL, token, f_kind, fp,
                                            planes.data_ptr() if planes is not None else None, N, 1 if keep else 0,
                                            fres.data_ptr() if want_fres else None, out4.data_ptr() if out4 is not None else None, ws.data_ptr(), wsb,
                                            _lib.stream_ptr(dev))
        _lib.check(rc, "nis_integrate_steps")
        if train:
            self._stats_bump(1)
        self._raise_on_status(ws[:4].view(torch.int32))
        return fres

    def rng_mode(self, G, N, train=True, f_kind=_lib.NIS_F_CAMEL, keep_planes=False):
        """How `integrate_steps_rng` draws for this call -- asked of the library itself (`nis_integrate_rng_mode`, the very
        predicate its launch planner uses): 2 = the d = 2 direct-row chain keeps float32 rows in a spare state plane, 1 = the
        ingest launch generates the float64 planes (every other model), 0 = invalid arguments."""
        return int(self.L.nis_integrate_rng_mode(self.handle, int(G), int(N), _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL,
                                                 int(f_kind), 1 if keep_planes else 0))

    def rng_supported(self, train=True, f_kind=_lib.NIS_F_CAMEL):
        """True when the first launch of a step can draw the batch itself: every model with a fusable built-in integrand."""
        return self.fusable() and self.rng_mode(1, 2, train, f_kind) > 0

    def integrate_step_bins(self, w, f_kind, f_params):
        """Test hook: one fused train-mode iteration on the production direct-row kernels, returning ((4,) moment record,
        (n_cells, N) int32 bin indices).  w: (N, 2) float32 CUDA tensor."""
        w = self._check_input(w, self.d, True)
        N, dev = w.shape[0], w.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev, 1)
        planes = self._planes_scratch(1, N, dev)
        bins = torch.full((len(self.cells), N), -1, dtype=torch.int32, device=dev)
        out4 = torch.zeros((1, 4), dtype=torch.double, device=dev)
        fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
        with _lib.on_device(dev):
            rc = self.L.nis_integrate_steps_bins(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64, N,
                                                 pp, rp, tp, self.pack_token(), f_kind, fp, planes.data_ptr(), N, bins.data_ptr(),
                                                 out4.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_integrate_steps_bins")
        self._stats_bump(1)
        return out4[0], bins & ~_lib.NIS_BIN_AMBIGUOUS_FLAG        # (FlowEngine.bin_ambiguous() tells whether any entry carried the flag)

    def integrate_steps_rng(self, G, N, dev, f_kind, f_params, out4, seed, stream0, token=None, w_out=None, rng_state=None,
                            train=True, planes=None, keep_planes=False, ws=None):
        """The same G iterations with the draw inside the step (manager.py:395): batch g is drawn on the device from Philox
        stream `stream0 + g` of key `seed` (mapping: include/nis_hip.h) by the first launch of the set; no input buffer exists.
        w_out: optional (G, N, d) float32 receiving the draws.  rng_state: optional int64[2] device tensor {seed, next stream}
        that overrides seed / stream0 and is advanced by G on the device (IntegratePlan: graph replays).
        planes: (G, d+1, N) float64 state to use (with keep_planes: X and J are left there); ws: workspace to use (a captured
        launch set owns both, so no other call can move or re-layout what its graph points at)."""
        _lib.require_gpu()
        dev = torch.device(dev)
        if train and N < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        pp, rp, tp = self._pointers(dev)
        if ws is None:
            ws, wsb = self._workspace(dev, G)
        else:
            wsb = self.L.nis_flow_workspace_bytes(self.handle, G)
            if ws.numel() < wsb:
                raise _lib.NisError("integrate_steps_rng: the workspace passed in is too small for this set size")
        if token is None:
            token = self.pack_token()
        if planes is None:
            planes = self._planes_scratch(G, N, dev)
        fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
        with _lib.on_device(dev):
            rc = self.L.nis_integrate_steps_rng(self.handle, G, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL, token,
                                                f_kind, fp, seed, stream0,
                                                rng_state.data_ptr() if rng_state is not None else None,
                                                w_out.data_ptr() if w_out is not None else None,
                                                planes.data_ptr() if planes is not None else None, N, 1 if keep_planes else 0,
                                                out4.data_ptr() if out4 is not None else None, ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_integrate_steps_rng")
        if train:
            self._stats_bump(1)
        if ws is self._ws:
            self._raise_on_status(ws[:4].view(torch.int32))
        return None

    def _planes_scratch(self, G, N, dev):
        """(G, d+1, N) float64 state planes kept between calls (nothing in them outlives a call when X / J are not kept)."""
        p = getattr(self, "_planes", None)
        if p is None or p.device != dev or p.shape[0] < G or p.shape[2] != N:
            self._planes = p = torch.empty((G, self.d + 1, N), dtype=torch.double, device=dev)
        return p[:G]

    def plan(self, G, N, dev, f_kind, f_params):
        """Cached IntegratePlan (one captured launch set) for this model state, set size and integrand."""
        for c in self.cells:
            ensure_flat(c, dev)
        # parameter VALUES (version counters) and the ADDRESSES the captured launches point at: parameter blob, BN running
        # statistics and counters (ensure_flat re-creates them after .to() / deepcopy).  The running statistics' values change
        # with every train-mode forward and do not enter a train-mode set.
        psig = tuple((p.data_ptr(), p._version) for c in self.cells for p in c.parameters()) + \
            tuple((c._flat.data_ptr(), c._running.data_ptr(), c._tracked.data_ptr()) for c in self.cells)
        key = (int(G), int(N), str(torch.device(dev)), int(f_kind), tuple(float(x) for x in f_params), psig)
        cache = getattr(self, "_plans", None)
        if cache is None or cache[0] != key:
            self._plans = cache = (key, IntegratePlan(self, G, N, dev, f_kind, f_params))
        return cache[1]

    def train_plan(self, G, N, dev, f, maxf, preburn, est):
        """Cached TrainPlan (two captured graphs) for this set size, integrand, loss form and the CURRENT addresses of the model's blobs."""
        for c in self.cells:
            ensure_flat(c, dev)
        key = (int(G), int(N), str(torch.device(dev)), int(f.nis_kind), tuple(float(x) for x in f.nis_params), bool(preburn), bool(est),
               maxf.data_ptr(), tuple((c._flat.data_ptr(), c._running.data_ptr(), c._tracked.data_ptr()) for c in self.cells))
        cache = getattr(self, "_train_plans", None)
        if cache is None:
            cache = self._train_plans = {}
        plan = cache.get(key)
        if plan is None:
            if len(cache) >= 4:                          # preburn / est / set-size variants of one training run; older ones are dropped
                cache.clear()
            plan = cache[key] = TrainPlan(self, G, N, dev, f, maxf, preburn, est)
        return plan

    integrate_step = integrate_steps


class IntegratePlan:
    """G iterations of integrate()'s loop body (draw -> flow -> camel * J -> moment records) as ONE hipGraph replay.

    Eagerly the 8 launches of a set reach the GPU one API call at a time and the first, short kernels of a set finish
    before the host has issued the next one: the first set after a synchronize costs ~90-330 us more than a later one
    (tools/graph_probe.py).  A captured set is submitted whole.  The generator state {seed, next stream} lives on the
    device and is advanced by the set's last launch, so every replay draws fresh numbers; records land in `self.out`.
    Valid while the model's parameters are unchanged (integrate() never changes them).

    The plan OWNS everything its graph points at besides the model's own blobs: workspace (parameter image, BN records and
    coefficients), state planes, generator state and output records.  No other engine call uses them, so nothing can be
    re-allocated, re-packed with another set size's layout or overwritten between replays."""

    def __init__(self, eng, G, N, dev, f_kind, f_params):
        self.eng, self.G, self.N, self.dev = eng, int(G), int(N), torch.device(dev)
        self.token = eng.pack_token()
        self.state = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.out = torch.zeros((self.G, 4), dtype=torch.double, device=self.dev)
        self.ws = torch.empty(eng.L.nis_flow_workspace_bytes(eng.handle, self.G), dtype=torch.uint8, device=self.dev)
        self.ws[:8].zero_()                              # status words (word 1 is sticky: FlowEngine.bin_ambiguous)
        self.planes = torch.empty((self.G, eng.d + 1, self.N), dtype=torch.double, device=self.dev)
        cells = eng.cells
        for c in cells:
            ensure_flat(c, self.dev)
        keep = [(c._running.clone(), c._tracked.clone()) for c in cells]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))

        def call(n):
            eng.integrate_steps_rng(self.G, n, self.dev, f_kind, f_params, self.out, 0, 0, token=self.token, rng_state=self.state,
                                    planes=self.planes, ws=self.ws)

        self._call = call

        with torch.cuda.stream(side):
            call(min(self.N, 512))                   # a small batch: loads the code objects, packs the parameter image, sizes the workspace
            side.synchronize()
            for c, (r, t) in zip(cells, keep):       # ... and must leave no trace in the BN running statistics / counters
                c._running.copy_(r)
                c._tracked.copy_(t)
            eng._stats_bump(-1)
            # thread_local: other threads of the process (the RCCL watchdog of a multi-rank job polls its events) must not
            # invalidate the capture.  If the capture fails anyway the plan launches eagerly -- same launches, same results.
            self.graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                    call(self.N)
                eng._stats_bump(-1)                  # capturing executed nothing
            except Exception as exc:                 # noqa: BLE001 - any capture failure: keep going without the graph
                import sys
                print(f"nf_amd: launch-set capture failed ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
                self.graph = None
        self.captured = self.graph is not None
        torch.cuda.current_stream(self.dev).wait_stream(side)

    def reset(self, seed, stream0=0):
        """Generator state for the next replay: batch g of it is Philox stream `stream0 + g` of key `seed`."""
        self.state.copy_(torch.tensor([seed, stream0], dtype=torch.int64))

    def run(self, dst, replay=None):
        """One set on the current stream; the (G, 4) records are copied to `dst`.  replay=None: the captured graph when the
        stream is idle (the set's first short kernels would otherwise wait for the host, ~90-330 us), eager launches when
        work is already queued (the host is ahead then, and back-to-back graph replays measured 20 % slower than eager
        launches: tools/plan_check.py).  Both forms run the same launches on the same device-resident generator state."""
        if replay is None:
            replay = torch.cuda.current_stream(self.dev).query()
        if replay and self.graph is not None:
            self.graph.replay()
            self.eng._stats_bump(1)
        else:
            self._call(self.N)
        dst.copy_(self.out)


class TrainPlan:
    """The library calls of ONE optimizer step -- G minibatches of N samples: forward with history + loss bookkeeping, and the adjoint
    seeded by the losses -- as TWO captured hipGraphs (replayed from `_HipEpochLossFn.forward` / `.backward`).

    In the reference's default regime (five minibatches of 2 000) the step is launch-bound: ~13 + ~16 dependent launches per narrow
    model (hundreds for a matrix-core one), each paying a host launch and an idle gap on the device.  A graph submits them whole.
    The plan OWNS every buffer its graphs point at (draws, history, BatchNorm records, integrand values, loss records, workspaces,
    gradient blobs); the model's own blobs (parameters, running statistics, counters) are read through their current addresses, which
    are part of the plan's key.  Parameter VALUES may change between replays: the parameter image is re-packed inside the graph.
    Built-in integrands only (a Python callable cannot be captured)."""

    def __init__(self, eng, G, N, dev, f, maxf, preburn, est):
        self.eng, self.G, self.N, self.dev = eng, int(G), int(N), torch.device(dev)
        L, d, n = eng.L, eng.d, len(eng.cells)
        self.preburn, self.est = bool(preburn), bool(est)
        dd, nset = d + 1, n + 1
        dbl = dict(dtype=torch.double, device=self.dev)
        self.w = torch.empty((self.G, self.N, d), **dbl)
        self.hist = torch.empty((self.G, nset, dd, self.N), **dbl)
        self.bnsave = torch.empty((self.G, L.nis_flow_bnsave_count(eng.handle)), **dbl)
        self.fv = torch.empty((self.G, self.N), **dbl)
        self.rec = torch.empty((self.G, 12), **dbl)
        self.out = torch.empty((self.G, 8), **dbl)
        self.gloss = torch.ones(1, **dbl)
        self.maxf = maxf                                   # device scalar owned by the training loop (its address is in the key)
        self.ws = torch.empty(L.nis_flow_workspace_bytes(eng.handle, self.G), dtype=torch.uint8, device=self.dev)
        self.lws = torch.empty(max(1, L.nis_loss_workspace_bytes_batched(self.G, self.N)), dtype=torch.uint8, device=self.dev)
        self.bws = torch.empty(L.nis_flow_backward_workspace_bytes_batched(eng.handle, self.G, self.N), dtype=torch.uint8, device=self.dev)
        for c in eng.cells:
            ensure_flat(c, self.dev)
        self.gflat = [torch.empty_like(c._flat) for c in eng.cells]
        self._gp = (C.c_void_p * n)(*[g.data_ptr() for g in self.gflat])
        self._fp = (C.c_double * 4)(*(list(f.nis_params) + [0.0] * 4)[:4])
        self._oc = (C.c_int32 * d)(*eng.out_col)
        self._kind = f.nis_kind
        set_ = dd * self.N
        self._x_planes = (self.hist.data_ptr() + 8 * (nset - 1) * set_, self.N, 1, nset * set_, self._oc)
        self._x_rows = (self.w.data_ptr(), 1, d, self.N * d, None)
        self._jac = (self.hist.data_ptr() + 8 * ((nset - 1) * set_ + d * self.N), 1, nset * set_)
        keep = [(c._running.clone(), c._tracked.clone()) for c in eng.cells]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        self.gA = self.gB = None
        with torch.cuda.stream(side):
            self.w.uniform_(0.05, 0.95)
            self._forward_calls()                      # warm-up: code objects, first pack; must leave no trace in the running statistics
            self._backward_calls()
            side.synchronize()
            for c, (r, t) in zip(eng.cells, keep):
                c._running.copy_(r)
                c._tracked.copy_(t)
            try:
                gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gA, stream=side, capture_error_mode="thread_local"):
                    self._forward_calls()
                with torch.cuda.graph(gB, stream=side, capture_error_mode="thread_local"):
                    self._backward_calls()
                self.gA, self.gB = gA, gB
            except Exception as exc:                   # noqa: BLE001 - any capture failure: the same calls run eagerly
                import sys
                print(f"nf_amd: training-step capture failed ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self.captured = self.gA is not None
        # ONE set of buffers: a second forward() overwrites the history the first one's backward still needs.  `generation` counts the
        # forwards, `_holder` is a weak reference to the token of the autograd node that owns the current history (dead once that node
        # has run its backward or was dropped): `busy()` tells the caller to take another path, `backward()` refuses a stale generation.
        self.generation = 0
        self._holder = None

    def busy(self):
        """True while the history of the last forward() is still owed to an autograd node that has not run its backward."""
        return self._holder is not None and self._holder() is not None

    def _forward_calls(self):
        eng, L = self.eng, self.eng.L
        pp, rp, tp = eng._pointers(self.dev)
        with _lib.on_device(self.dev):
            st = _lib.stream_ptr(self.dev)
            rc = L.nis_flow_forward_hist_batched(eng.handle, self.w.data_ptr(), _lib.NIS_F64, self.G, self.N, pp, rp, tp, _lib.NIS_BN_TRAIN,
                                                 0, self.hist.data_ptr(), self.N, self.bnsave.data_ptr(), None, self.ws.data_ptr(),
                                                 self.ws.numel(), st)          # pack_token 0: the parameter image is re-packed on every replay
            _lib.check(rc, "nis_flow_forward_hist_batched")
            src = self._x_rows if self.preburn else self._x_planes
            rc = L.nis_loss_terms_batched(None, *src, eng.d, self._kind, self._fp, self.fv.data_ptr(), *self._jac, self.G, self.N,
                                          self.maxf.data_ptr(), 1 if self.preburn else 0, 1 if self.est else 0, self.rec.data_ptr(),
                                          self.out.data_ptr(), self.lws.data_ptr(), self.lws.numel(), st)
            _lib.check(rc, "nis_loss_terms_batched")

    def _backward_calls(self):
        eng, L = self.eng, self.eng.L
        pp, _, _ = eng._pointers(self.dev)
        with _lib.on_device(self.dev):
            rc = L.nis_flow_backward_loss_batched(eng.handle, self.hist.data_ptr(), self.N, self.G, self.N, pp, self.bnsave.data_ptr(),
                                                  _lib.NIS_BN_TRAIN, self.fv.data_ptr(), self.maxf.data_ptr(), self.rec.data_ptr(),
                                                  self.gloss.data_ptr(), 1 if self.est else 0, self._gp, self.bws.data_ptr(),
                                                  self.bws.numel(), _lib.stream_ptr(self.dev))
            _lib.check(rc, "nis_flow_backward_loss_batched")

    def forward(self, w, holder=None):
        """Draws (G, N, d) -> bookkeeping rows (G, 8) (a fresh tensor; the plan's own copy is overwritten by the next replay).
        `holder`: any weak-referenceable object that lives as long as the caller still wants backward() for THIS forward."""
        if self.busy():
            raise _lib.NisError("TrainPlan.forward: the previous forward's history is still awaiting its backward (check busy() and take "
                                "the eager path or another plan)")
        import weakref
        self.generation += 1
        self._holder = weakref.ref(holder) if holder is not None else None
        self.w.copy_(w)
        if self.gA is not None:
            self.gA.replay()
        else:
            self._forward_calls()
        self.eng._stats_bump(1)
        self.eng._raise_on_status(self.ws[:4].view(torch.int32))
        return self.out.clone()

    def backward(self, gloss, generation=None):
        """Upstream cotangent of the summed losses -> [gradient blob per cell] (fresh tensors).  `generation`: the value of
        `self.generation` right after the forward this backward belongs to; a later forward has overwritten the history -> error."""
        if generation is not None and generation != self.generation:
            raise _lib.NisError(f"TrainPlan.backward: stale history (forward #{generation}, the plan is at #{self.generation}): the gradient "
                                "would be that of another minibatch set")
        self.gloss.copy_(gloss.detach().reshape(1))
        if self.gB is not None:
            self.gB.replay()
        else:
            self._backward_calls()
        return [g.clone() for g in self.gflat]


def philox_rows(seed, stream0, G, N, d, dev):
    """(G, N, d) float32 uniform draws of the library's counter-based generator: exactly the numbers `integrate_steps_rng`
    draws inside its first launch for the same seed / streams (include/nis_hip.h gives the mapping for d == 2 and d != 2)."""
    _lib.require_gpu()
    dev = torch.device(dev)
    out = torch.empty((G, N, d), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nis_philox_rows(seed, stream0, G, N, d, out.data_ptr(), _lib.stream_ptr(dev)), "nis_philox_rows")
    return out


def moments_of(planes, d, out_col, f_kind, f_params, want_fres=False):
    """f(X)*J from the planes + (count, mean, M2, max) on device.  Returns (out4 tensor, fres or None)."""
    L = _lib.lib()
    N = planes.shape[1]
    dev = planes.device
    out4 = torch.empty(4, dtype=torch.double, device=dev)
    wsb = L.nis_moments_workspace_bytes()
    ws = _moments_ws(dev, wsb)
    fres = torch.empty(N, dtype=torch.double, device=dev) if want_fres else None
    oc = (C.c_int32 * d)(*out_col)
    fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
    with _lib.on_device(dev):
        rc = L.nis_integrand_moments(planes.data_ptr(), N, N, d, oc, f_kind, fp, fres.data_ptr() if want_fres else None,
                                     out4.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "nis_integrand_moments")
    return out4, fres


_MOM_WS = {}


def _moments_ws(dev, nbytes):
    """Per-device scratch of the reduction kernels (per-workgroup records), kept between calls: the merge launch that reads it
    is queued right behind the launch that writes it, on the same stream."""
    key = (dev.type, dev.index or 0)
    ws = _MOM_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _MOM_WS[key] = ws = torch.empty(max(nbytes, 1 << 18), dtype=torch.uint8, device=dev)
    return ws


def moments(a, b=None):
    """(count, mean, M2, max) of a (or a*b) as a 4-element device tensor: the HIP replacement of the
    torch.mean / torch.var / torch.max passes at manager.py:149-165, :241-268, :399-400."""
    _lib.require_gpu()
    L = _lib.lib()
    a = a.detach().to(torch.double).contiguous()
    if b is not None:
        b = b.detach().to(torch.double).contiguous()
    out4 = torch.empty(4, dtype=torch.double, device=a.device)
    wsb = L.nis_moments_workspace_bytes()
    ws = _moments_ws(a.device, wsb)
    with _lib.on_device(a.device):
        rc = L.nis_moments(a.data_ptr(), b.data_ptr() if b is not None else None, a.numel(), out4.data_ptr(),
                           ws.data_ptr(), wsb, _lib.stream_ptr(a.device))
    _lib.check(rc, "nis_moments")
    return out4
