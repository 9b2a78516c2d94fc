"""Over-read / over-write probe: every caller-owned buffer of the main entry points is a view at the very END of its own 16 MiB
allocation (an exact-size allocator segment), so an access past the buffer's last element leaves the segment -- small and ragged batch
sizes, where prefetching kernels are most likely to run off the end.  A memory fault aborts the process: tests run this script in a
subprocess and look for GUARD_OK.   python tools/guard_check.py [1|2]     2: torch.empty itself hands out segment-end views, so the
buffers the Python layer allocates (workspaces, state planes, history, records, gradient blobs) are covered as well."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import _lib, integrands
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager
from nf_amd.normalizing_flows._autograd import flow_loss_batched
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
from nf_amd.flow import moments

dev = torch.device("cuda:0")
SEG = 16 << 20
_keep = []


def tail(shape, dtype, fill=None):
    """A tensor of `shape` whose last byte is the last byte of a fresh 16 MiB allocation."""
    n = 1
    for s in shape:
        n *= s
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    pad = (-nbytes) % 16                                   # keep the start 16-byte aligned (the vector-load paths check alignment)
    big = _orig_empty(SEG, dtype=torch.uint8, device=dev)
    _keep.append(big)
    t = big[SEG - nbytes - pad:SEG - pad].view(dtype).view(shape) if pad else big[SEG - nbytes:].view(dtype).view(shape)
    if fill == "rand":
        t.copy_(torch.rand(shape, device=dev).to(dtype) * (1 - 2 ** -24))
    elif fill is not None:
        t.fill_(fill)
    return t


def sync(msg):
    torch.cuda.synchronize()
    del _keep[:]
    print("ok", msg, flush=True)


_orig_empty = torch.empty


def guarded_empty(*size, **kw):
    """torch.empty for CUDA tensors below 4 MiB as a view at the end of its own 16 MiB segment: the library-owned buffers of the
    Python layer (workspaces, state planes, history, records, gradient blobs) get the same treatment as the caller's."""
    dev_ = kw.get("device")
    shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
    dtype = kw.get("dtype") or torch.get_default_dtype()
    if dev_ is None or torch.device(dev_).type != "cuda" or any(k not in ("device", "dtype") for k in kw):
        return _orig_empty(*size, **kw)
    n = 1
    for s_ in shape:
        n *= int(s_)
    nbytes = n * _orig_empty((), dtype=dtype).element_size()
    if nbytes == 0 or nbytes > (4 << 20):
        return _orig_empty(*size, **kw)
    pad = (-nbytes) % 16
    big = _orig_empty(SEG, dtype=torch.uint8, device=dev_)
    return big[SEG - nbytes - pad:SEG - pad].view(dtype).view(shape)


PASS = 1 if len(sys.argv) < 2 else int(sys.argv[1])
if PASS == 2:
    torch.empty = guarded_empty
torch.manual_seed(0)
sizes = (2, 3, 5, 63, 64, 65, 127, 257) if PASS == 1 else (2, 3, 65, 130)
f = integrands.Camel()
# 1 + 2: the d = 2 SPEC class, fed draws at a segment end; kept planes at a segment end (caller-side integrand form)
NF = PWQuadManager(n_flow=2); NF.create_model(2, 4, [3] * 3)
eng = NF.best_model.engine()
for train in (True, False):
    NF.best_model.train(train)
    for N in sizes:
        for G in (1, 3):
            w = tail((G, N, 2), torch.float32, "rand")
            out = tail((G, 4), torch.double, 0.0)
            eng.integrate_steps(w, train, f.nis_kind, f.nis_params, out)
            planes = tail((G, 3, N), torch.double, 0.0)
            eng.integrate_steps(w, train, _lib.NIS_F_NONE, [], None, planes=planes)
            eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 7, 0, train=train)
            eng.integrate_steps_rng(G, N, dev, _lib.NIS_F_NONE, [], None, 7, 0, train=train, planes=planes, keep_planes=True)
    sync(f"C2 class train={train}")
# 3: generic narrow and matrix-core models, forward + fused integrate
for mk, d in ((lambda: (lambda m: (m.create_model(4, 6, [8, 8]), m)[1])(PWQuadManager(n_flow=5)), 5),
              (lambda: (lambda m: (m.create_model(3, 8, [16, 16]), m)[1])(PWQuadManager(n_flow=6)), 6),
              (lambda: (lambda m: (m.create_model(2, 3, 40, [32, 32], 2), m)[1])(PWLinManager(n_flow=4)), 4)):
    M = mk()
    e = M.best_model.engine()
    fn = integrands.Normal(0.5, 0.3, d)
    for train in (True, False):
        M.best_model.train(train)
        for N in sizes:
            w = tail((N, d), torch.double, "rand")
            e.forward(w, False, train, want_xj=True, want_bins=True)
            wg = tail((2, N, d), torch.float32, "rand")
            out = tail((2, 4), torch.double, 0.0)
            e.integrate_steps(wg, train, fn.nis_kind, fn.nis_params, out)
    sync(f"model d={d}")
    # 5: the fused training step (history / records are library-owned; the draws are the caller's)
    M.best_model.train(True)
    maxf = torch.ones((), dtype=torch.double, device=dev)
    for N in (2, 3, 65, 130):
        w = tail((3, N, d), torch.double, "rand")
        ls, _ = flow_loss_batched(M.best_model, w, fn, maxf, False, False)
        (ls / 3).backward()
    sync(f"training d={d}")
# 6: RAMBO, rows at a segment end, outputs at segment ends
gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
for N in (1, 2, 3, 127, 128, 129, 257):
    r = tail((N, 8), torch.double, "rand")
    gen.generateKinematics_batch(1000., r, pT_mincut=0, delR_mincut=0, rap_maxcut=-1, pdgs=[2, -1])
gen2 = FlatInvertiblePhasespace([0., 0.], [0.] * 6, pdf=None, pdf_active=True)
for N in (1, 3, 129):
    r = tail((N, 16), torch.double, "rand")
    gen2.generateKinematics_batch(13000., r, pT_mincut=20, delR_mincut=0.4, rap_maxcut=2.5, pdgs=[21, 21])
sync("rambo")
# 7: reductions
for N in (1, 2, 63, 65, 1025):
    a = tail((N,), torch.double, "rand")
    b = tail((N,), torch.double, "rand")
    moments(a, b)
sync("moments")
print("GUARD_OK")
