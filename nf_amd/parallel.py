"""Multi-GPU sampling: one process per GPU, samples sharded, ONE collective to combine the estimators.

The reference has no distributed code; this is the MI355X-native scaling path the
north star asks for: independent sample shards on the 8 GPUs of a node and a single RCCL all-reduce
over xGMI of the per-rank moment records.  Every rank writes its (nitn, 4) records
{count, mean, M2, max} into its own slot of a zero-initialised (world, nitn, 4) buffer; the all-reduce
(SUM) of that buffer is an exact gather (x + 0 is exact), after which every rank merges the slots in
rank order with Chan's update -- deterministic and free of the cancellation a sum of squares has.
Payload: world * nitn * 32 bytes, latency bound.

Each shard is its own train-mode BatchNorm batch ("independent batches", SURVEY.md 8(e) option 2): it is
parity-checked per shard; shards of an eval-mode model reproduce the single-batch result exactly.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def merge_moments(recs):
    """Merge moment records along axis -2: recs (..., R, 4) -> (..., 4).  NumPy or torch input."""
    is_t = torch.is_tensor(recs)
    a = recs.detach().cpu().numpy() if is_t else np.asarray(recs, dtype=np.float64)
    out = np.zeros(a.shape[:-2] + (4,))
    out[..., 3] = -np.inf
    for r in range(a.shape[-2]):
        n_b, mean_b, m2_b, mx_b = a[..., r, 0], a[..., r, 1], a[..., r, 2], a[..., r, 3]
        n_a, mean_a, m2_a = out[..., 0].copy(), out[..., 1].copy(), out[..., 2].copy()
        n = n_a + n_b
        with np.errstate(divide="ignore", invalid="ignore"):
            fb = np.where(n > 0, n_b / np.where(n > 0, n, 1), 0.0)
        delta = mean_b - mean_a
        out[..., 0] = n
        out[..., 1] = np.where(n_b > 0, mean_a + delta * fb, mean_a)
        out[..., 2] = np.where(n_b > 0, m2_a + m2_b + delta * delta * (n_a * fb), m2_a)
        out[..., 3] = np.where(n_b > 0, np.maximum(out[..., 3], mx_b), out[..., 3])
    return torch.as_tensor(out) if is_t else out


def merge_moments_t(recs):
    """The same merge along axis -2 in torch ops, on whatever device `recs` lives (no host round trip, no Python loop over the
    ranks): with n = sum n_r and mean = sum n_r mean_r / n,  M2 = sum M2_r + sum n_r (mean_r - mean)^2  -- every term is
    non-negative, so no cancellation either; equal to the sequential Chan merge to rounding (1e-15).  Empty records (n_r == 0)
    contribute nothing, whatever their mean / max slots hold."""
    n_r, mean_r, m2_r, mx_r = recs[..., 0], recs[..., 1], recs[..., 2], recs[..., 3]
    has = n_r > 0
    n = n_r.sum(-1)
    safe = torch.where(n > 0, n, torch.ones_like(n))
    mean = (n_r * torch.where(has, mean_r, torch.zeros_like(mean_r))).sum(-1) / safe
    d = torch.where(has, mean_r - mean.unsqueeze(-1), torch.zeros_like(mean_r))
    m2 = torch.where(has, m2_r, torch.zeros_like(m2_r)).sum(-1) + (n_r * d * d).sum(-1)
    mx = torch.where(has, mx_r, torch.full_like(mx_r, float("-inf"))).amax(-1)
    return torch.stack([n, mean, m2, mx], dim=-1)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None, timeout_s=300):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); backend "nccl" (= RCCL)
    on GPUs, "gloo" on CPU.  Returns (rank, world_size, local_rank).

    RCCL needs one GPU per rank: the call fails fast with a clear message when the node shows fewer devices than ranks
    (two ranks on one device end in `ncclInvalidUsage: Duplicate GPU detected` deep inside the first collective otherwise),
    binds the communicator to this rank's device (`device_id`: no "guessing device ID" warning, no lazy-init hang) and
    sets an explicit timeout.  `NIS_DIST_BACKEND=gloo` rehearses several ranks on one GPU."""
    import datetime
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("NIS_DIST_BACKEND") or backend        # e.g. gloo to rehearse N ranks on one GPU
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("NIS_DIST_TIMEOUT_S", timeout_s)))}
        if torch.cuda.is_available():
            ndev = torch.cuda.device_count()
            local_ws = int(os.environ.get("LOCAL_WORLD_SIZE", ws))
            if backend == "nccl" and ndev < local_ws:
                raise RuntimeError(f"RCCL needs one GPU per rank: {local_ws} local ranks but {ndev} visible device(s) "
                                   "(set NIS_DIST_BACKEND=gloo to rehearse several ranks on one GPU)")
            torch.cuda.set_device(local % ndev)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local % ndev)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws, **kw)
    return rank, ws, local


def merge_gathered(buf):
    """(world, nitn, 4) gathered records on a GPU -> (nitn, 4), one launch (`nis_merge_moments`)."""
    from . import _lib
    ws = buf.shape[0]
    out = torch.empty(tuple(buf.shape[1:]), dtype=torch.double, device=buf.device)
    steps = out.numel() // 4
    with _lib.on_device(buf.device):
        _lib.check(_lib.lib().nis_merge_moments(buf.data_ptr(), ws, steps, out.data_ptr(), _lib.stream_ptr(buf.device)),
                   "nis_merge_moments")
    return out


def allreduce_moments(local, group=None):
    """local: (nitn, 4) float64 records of this rank (any device).  Returns the merged (nitn, 4) records, identical on every rank.

    ONE collective on every backend and device -- the single all-reduce SURVEY.md 8(e) / the north star name: all-reduce (SUM) of a
    zero-padded (world, nitn, 4) buffer in which a rank fills only its own slot.  x + 0 is exact, so the result is an exact gather
    (world x nitn x 32 bytes: latency-bound), and every rank then applies the same Chan merge to identical inputs: on a GPU ONE launch
    (`nis_merge_moments`; collective and merge are inside the job's timed region), on CPU tensors (gloo tests without a GPU) the same
    formula in torch ops.  RCCL and the gloo rehearsal of several ranks on one GPU run the same branch; a failing collective raises."""
    rank, ws = world()
    if ws == 1:
        return local
    buf = torch.zeros((ws,) + tuple(local.shape), dtype=torch.double, device=local.device)
    buf[rank] = local
    # max slot: an empty rank holds -inf there; kept out of the SUM (-inf + 0 is fine, but a NaN-free buffer is the simpler contract)
    buf[rank, ..., 3] = torch.nan_to_num(local[..., 3], neginf=0.0)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if local.is_cuda:
        return merge_gathered(buf)                  # (world, nitn, 4) -> (nitn, 4), one launch
    recs = buf.movedim(0, -2)                       # (nitn, world, 4)
    return merge_moments_t(recs)


def shard_size(total, rank, ws):
    """Contiguous split of `total` samples: the first total % ws ranks take one extra."""
    base, rem = divmod(int(total), ws)
    return base + (1 if rank < rem else 0)


def moments_batches_syncbn(manager, f, nitn, n_local, dev):
    """Per-iteration moment records of this rank's shard with EXACT batch statistics (SURVEY.md 8(e) option 1): every
    BatchNorm layer sees the statistics of the whole sharded batch (`FlowEngine.forward_sync`), so the shards together
    reproduce the reference's single-batch iteration.  One small all-gather per BatchNorm layer and cell."""
    from .flow import moments as hip_moments, moments_of
    model = manager.best_model
    model.to(dev)
    eng = model.engine()
    out = torch.empty((nitn, 4), dtype=torch.double, device=dev)
    for i in range(nitn):
        w = manager._uniform(n_local, dev, torch.float32, "integrate")
        res = eng.forward_sync(w, want_xj=not hasattr(f, "nis_kind"))
        if hasattr(f, "nis_kind"):
            out[i] = moments_of(res["planes"], eng.d, eng.out_col, f.nis_kind, f.nis_params)[0]
        else:
            XJ = res["xj"]
            out[i] = hip_moments(f(XJ[:, :-1]), XJ[:, -1])
    return out


def integrate_sharded(manager, f, nitn, neval, dev=None, sync_bn=False):
    """`BasicManager.integrate` with the neval points of every iteration split across the ranks of the
    default process group.  Same return convention (two 0-d float32 CPU tensors), same on every rank.
    sync_bn=False: every shard is its own train-mode BatchNorm batch (independent batches, no collective on the data
    path); sync_bn=True: exact whole-batch statistics (`moments_batches_syncbn`)."""
    rank, ws = world()
    if manager.best_model is None:
        print("No model has been trained")
        return (0, 0)
    dev = manager._device(dev if dev is not None else manager._model_device())
    n_local = shard_size(neval, rank, ws)
    if sync_bn and ws > 1 and manager.best_model.training:
        local = moments_batches_syncbn(manager, f, int(nitn), n_local, dev)
    else:
        local = manager.moments_batches(f, int(nitn), n_local, dev)
    mom = allreduce_moments(local).cpu()
    var = (mom[:, 2] / (mom[:, 0] - 1.0)).to(torch.float32)
    mean = mom[:, 1].to(torch.float32)
    sig = torch.sum(mean / var) / torch.sum(1 / var)
    sig_err = torch.sqrt(1 / torch.sum(1 / var)) / np.sqrt(int(neval) * int(nitn))
    return (sig, sig_err)


class DataParallel:
    """Sample-sharded training ("data-parallel - same sharding + gradient all-reduce").

    Every rank runs `_train_variance_forward_seq` on its OWN minibatches (`batch_size` / `mini_batch_size` are per
    rank; seed the ranks differently), as independent train-mode BatchNorm batches.  Per optimizer step ONE all-reduce
    carries the flattened parameter gradients (<= 448 KB at C5) and the bookkeeping scalars (loss, variance, estimator
    terms), averaged over the ranks, so the optimizer states and the early-stopping decisions stay identical everywhere.
    The BatchNorm running statistics (every rank's train-mode forwards advance them from its OWN minibatches) are merged
    whenever a copy of the model is kept (`sync_buffers`: pooled mean / variance), so `best_model`'s state_dict, checkpoints
    and the eval-mode `integrate=True` tail are identical on every rank.  `max f` of the survey is combined with one
    all-reduce(MAX), and the survey's loss / variance are normalised with that GLOBAL maximum.

        rank, ws, _ = parallel.init_from_env()
        NF = PWQuadManager(n_flow=2); NF.create_model(2, 4, [3] * 3)
        parallel.DataParallel(NF)                      # broadcasts rank 0's parameters / BN buffers, installs itself
        NF._train_variance_forward_seq(f, optim, ...)
    """

    def __init__(self, manager, group=None, src=0):
        self.group = group
        self.rank, self.world = world()
        self.manager = manager
        if self.world > 1:
            with torch.no_grad():
                for t in list(manager._model.parameters()) + list(manager._model.buffers()):
                    dist.broadcast(t, src=src, group=group)
        manager.data_parallel = self

    def sync_max(self, t):
        """Element-wise maximum over the ranks (survey: max |f|)."""
        if self.world == 1:
            return t
        t = t.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def mean(self, t):
        """Average of a tensor over the ranks."""
        if self.world == 1:
            return t
        t = t.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world

    def sync_buffers(self, model):
        """Merge the BatchNorm running statistics of `model` over the ranks (one all-reduce): called before a copy of the model
        is kept (`best_model`), so that what is saved / evaluated later is the same on every rank.  running_mean is averaged;
        running_var is merged like the variance of pooled groups -- mean of the ranks' variances PLUS the variance of their
        means -- so the buffers match what a single process over the union of the minibatches would hold, not a value that
        systematically lacks the between-rank spread of the means."""
        if self.world == 1:
            return
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d) and m.running_mean is not None]
        if not bns:
            return
        mean = torch.cat([m.running_mean.detach().reshape(-1).double() for m in bns])
        var = torch.cat([m.running_var.detach().reshape(-1).double() for m in bns])
        flat = torch.cat([mean, var + mean * mean])                 # E[x], E[x^2] of the pooled running estimate
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat /= self.world
        n = mean.numel()
        gmean, gvar = flat[:n], (flat[n:] - flat[:n] * flat[:n]).clamp_min(0.0)
        off = 0
        with torch.no_grad():
            for m in bns:
                k = m.running_mean.numel()
                m.running_mean.copy_(gmean[off:off + k].view_as(m.running_mean))
                m.running_var.copy_(gvar[off:off + k].view_as(m.running_var))
                off += k

    def sync_step(self, model, scalars):
        """Average the gradients of `model` (in place) and the 1-d tensor `scalars` over the ranks: one all-reduce.  The BatchNorm
        running statistics are NOT part of it: they do not enter a train-mode step, and touching them every step would bump their
        version counters (re-packing the kernels' parameter image each minibatch); `sync_buffers` merges them whenever a copy of
        the model is kept."""
        if self.world == 1:
            return scalars
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads] + [scalars.detach().reshape(-1).to(grads[0].dtype)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat /= self.world
        off = 0
        with torch.no_grad():
            for g in grads:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        return flat[off:].to(scalars.dtype)
