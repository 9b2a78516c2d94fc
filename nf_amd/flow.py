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
    bns = [m for m in cell.NN if isinstance(m, torch.nn.BatchNorm1d)]
    lins = [m for m in cell.NN if isinstance(m, torch.nn.Linear)]
    ps, rs = [], []
    for l, bn in enumerate(bns):
        ps += [bn.weight, bn.bias, lins[l].weight]
        if l < len(bns) - 1 and lins[l].bias is not None:       # AffineCoupling: biased hidden Linears
            ps.append(lins[l].bias)
        rs += [(bn, "running_mean"), (bn, "running_var")]
    ps.append(lins[-1].bias)
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
        self._stats_epoch = 0

    def fusable(self):
        """The built-in-integrand rider lives in the narrow (register width <= 8) apply kernels."""
        last = self.cells[-1]
        return max([last.pass_through_size] + list(last.hidden_widths)) <= 8

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
            self._sig = None
        return self._ws, wsb

    def pack_token(self):
        """Changes whenever a parameter / BN buffer was modified (torch version counters) or a train-mode
        forward of ours advanced the running statistics: libnis_hip then re-packs its parameter image."""
        sig = (self._stats_epoch,) + tuple(t._version for c in self.cells for t in c.state_dict(keep_vars=True).values())
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
        rc = self.L.nis_flow_forward(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                     1 if has_jac else 0, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL,
                                     token, planes.data_ptr(), N, xj.data_ptr() if want_xj else None,
                                     bins.data_ptr() if want_bins else None, ws.data_ptr(), wsb, _lib.stream_ptr())
        _lib.check(rc, "nis_flow_forward")
        if train:
            self._stats_epoch += 1
        status = ws[:4].view(torch.int32)
        self._raise_on_status(status)
        return dict(planes=planes, xj=xj, bins=bins, status=status, ws=ws)

    def forward_sync(self, w, has_jac=False, want_xj=True, group=None):
        """Train-mode forward of THIS RANK'S rows of a batch that is sharded over the ranks of `group`, with the BatchNorm
        statistics of the whole batch (SURVEY.md 8(e) option 1): `nis_flow_forward_sync`, whose exchange hook is one
        in-place all-gather (RCCL on GPUs) of `world` x (3 x width) doubles per BatchNorm layer.  Returns the same dict as
        `forward`; concatenating the ranks' results reproduces the single-process forward of the concatenated batch."""
        import torch.distributed as dist
        w = self._check_input(w, self.d + (1 if has_jac else 0), False)      # a shard may hold a single row
        N, dev = w.shape[0], w.device
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev)
        planes = torch.empty((self.d + 1, N), dtype=torch.double, device=dev)
        xj = torch.empty((N, self.d + 1), dtype=torch.double, device=dev) if want_xj else None
        buf = torch.zeros(world * 64 * 3, dtype=torch.double, device=dev)

        failure = []

        def exchange(ctx, records, count, stream):
            try:                                             # an exception must not vanish inside the ctypes callback
                flat = buf[:world * count]
                dist.all_gather_into_tensor(flat, flat[rank * count:(rank + 1) * count].clone(), group=group)
            except BaseException as exc:                     # noqa: BLE001 - re-raised after the library call returns
                failure.append(exc)

        hook = _lib.SYNC_FN(exchange)
        rc = self.L.nis_flow_forward_sync(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                          1 if has_jac else 0, N, pp, rp, tp, self.pack_token(), planes.data_ptr(), N,
                                          xj.data_ptr() if want_xj else None, ws.data_ptr(), wsb, _lib.stream_ptr(), hook, None,
                                          buf.data_ptr(), world, rank)
        if failure:
            raise failure[0]
        _lib.check(rc, "nis_flow_forward_sync")
        self._stats_epoch += 1
        status = ws[:4].view(torch.int32)
        self._raise_on_status(status)
        return dict(planes=planes, xj=xj, bins=None, status=status, ws=ws)

    def _raise_on_status(self, status):
        """PWLin with x == 1.0 indexes bin n_bins: the reference's torch.gather raises (coupling_cells.py:136); the kernel
        flags it instead.  Only models with a PWLin cell pay the device->host read of the status word."""
        if any(c.KIND == _lib.NIS_PWLIN for c in self.cells) and int(status[0].item()) & 1:
            raise IndexError("index out of range in PWLin bin lookup (an input coordinate equals 1.0 or is negative)")

    # ---- training: forward with history + hand-written adjoint
    def backward_supported(self):
        """True when every cell is in the class the adjoint kernels cover (include/nis_hip.h)."""
        return bool(self.L.nis_flow_backward_supported(self.handle))

    def forward_train(self, w, has_jac):
        """Train-mode forward that keeps what the adjoint needs.  Returns (xj, saved)."""
        w = self._check_input(w, self.d + (1 if has_jac else 0), True)
        N, dev = w.shape[0], w.device
        pp, rp, tp = self._pointers(dev)
        ws, wsb = self._workspace(dev)
        hist = torch.empty((len(self.cells) + 1, self.d + 1, N), dtype=torch.double, device=dev)
        bnsave = torch.empty(self.L.nis_flow_bnsave_count(self.handle), dtype=torch.double, device=dev)
        xj = torch.empty((N, self.d + 1), dtype=torch.double, device=dev)
        rc = self.L.nis_flow_forward_train(self.handle, w.data_ptr(), _lib.NIS_F32 if w.dtype == torch.float32 else _lib.NIS_F64,
                                           1 if has_jac else 0, N, pp, rp, tp, self.pack_token(), hist.data_ptr(), N,
                                           bnsave.data_ptr(), xj.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr())
        _lib.check(rc, "nis_flow_forward_train")
        self._stats_epoch += 1
        self._raise_on_status(ws[:4].view(torch.int32))
        return xj, (hist, bnsave, has_jac)

    def backward(self, saved, gxj, want_gx):
        """dL/dXJ (N, d+1) -> ([flat parameter-gradient blob per cell], dL/dx or None)."""
        hist, bnsave, has_jac = saved
        N, dev = hist.shape[2], hist.device
        pp, _, _ = self._pointers(dev)
        n = len(self.cells)
        gflat = [torch.empty_like(c._flat) for c in self.cells]
        gp = (C.c_void_p * n)(*[g.data_ptr() for g in gflat])
        wsb = self.L.nis_flow_backward_workspace_bytes(self.handle, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        gx = torch.empty((N, self.d + 1), dtype=torch.double, device=dev) if want_gx else None
        gxj = gxj.to(torch.double).contiguous()
        rc = self.L.nis_flow_backward(self.handle, hist.data_ptr(), N, N, 1 if has_jac else 0, pp, bnsave.data_ptr(), gxj.data_ptr(),
                                      gp, gx.data_ptr() if want_gx else None, ws.data_ptr(), wsb, _lib.stream_ptr())
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
            planes = torch.empty((G, self.d + 1, N), dtype=torch.double, device=dev)
        fres = torch.empty((G, N), dtype=torch.double, device=dev) if want_fres else None
        fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
        rc = self.L.nis_integrate_steps(self.handle, w2.data_ptr(), _lib.NIS_F32 if w2.dtype == torch.float32 else _lib.NIS_F64,
                                        G, N, pp, rp, tp, _lib.NIS_BN_TRAIN if train else _lib.NIS_BN_EVAL, token, f_kind, fp,
                                        planes.data_ptr(), N, 1 if keep else 0, fres.data_ptr() if want_fres else None,
                                        out4.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr())
        _lib.check(rc, "nis_integrate_steps")
        if train:
            self._stats_epoch += 1
        self._raise_on_status(ws[:4].view(torch.int32))
        return fres

    integrate_step = integrate_steps


def moments_of(planes, d, out_col, f_kind, f_params, want_fres=False):
    """f(X)*J from the planes + (count, mean, M2, max) on device.  Returns (out4 tensor, fres or None)."""
    L = _lib.lib()
    N = planes.shape[1]
    dev = planes.device
    out4 = torch.empty(4, dtype=torch.double, device=dev)
    wsb = L.nis_moments_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    fres = torch.empty(N, dtype=torch.double, device=dev) if want_fres else None
    oc = (C.c_int32 * d)(*out_col)
    fp = (C.c_double * 4)(*(list(f_params) + [0.0] * 4)[:4])
    rc = L.nis_integrand_moments(planes.data_ptr(), N, N, d, oc, f_kind, fp, fres.data_ptr() if want_fres else None,
                                 out4.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr())
    _lib.check(rc, "nis_integrand_moments")
    return out4, fres


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
    ws = torch.empty(wsb, dtype=torch.uint8, device=a.device)
    rc = L.nis_moments(a.data_ptr(), b.data_ptr() if b is not None else None, a.numel(), out4.data_ptr(),
                       ws.data_ptr(), wsb, _lib.stream_ptr())
    _lib.check(rc, "nis_moments")
    return out4
