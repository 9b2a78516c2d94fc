"""N>1 path on CPU: two `gloo` ranks shard a batch, build their moment records with the oracle and
combine them through the SAME `allreduce_moments` the RCCL path uses; the result must equal the
single-batch estimators.  (Per-rank records come from the oracle here because the HIP kernels need a GPU;
on the GPU box tests/test_gpu_parity.py covers kernel records + merge.)"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nf_amd.parallel import DataParallel, allreduce_moments, merge_moments, shard_size
from oracle import flow as oflow


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(v):
    n, m, var, mx = oflow.moments(v)
    return [n, m, var * (n - 1), mx]


def _worker(rank, ws, port, outdir, neval, nitn):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rng = np.random.default_rng(42)
    recs = []
    for it in range(nitn):
        x = rng.random((neval, 2))                      # every rank draws the full batch, keeps its slice
        lo = sum(shard_size(neval, r, ws) for r in range(rank))
        hi = lo + shard_size(neval, rank, ws)
        recs.append(_records(oflow.camel(x[lo:hi]) * (1.0 + 0.1 * x[lo:hi, 0])))
    local = torch.tensor(recs, dtype=torch.double)
    merged = allreduce_moments(local)
    np.save(os.path.join(outdir, f"r{rank}.npy"), merged.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("ws,neval", [(2, 10001), (2, 4096)])
def test_two_rank_allreduce_equals_single_batch(ws, neval):
    nitn = 3
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(ws, port, td, neval, nitn), nprocs=ws, join=True)
        outs = [np.load(os.path.join(td, f"r{r}.npy")) for r in range(ws)]
    np.testing.assert_array_equal(outs[0], outs[1])         # identical on every rank
    rng = np.random.default_rng(42)
    for it in range(nitn):
        x = rng.random((neval, 2))
        ref = _records(oflow.camel(x) * (1.0 + 0.1 * x[:, 0]))
        np.testing.assert_allclose(outs[0][it, :3], ref[:3], rtol=1e-12)
        assert outs[0][it, 3] == ref[3]


def test_merge_moments_handles_empty_and_uneven_shards():
    rng = np.random.default_rng(0)
    v = rng.normal(5.0, 1e-3, 1000)
    cuts = [0, 0, 17, 17, 600, 1000]
    recs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        recs.append(_records(v[a:b]) if b > a else [0.0, 0.0, 0.0, -np.inf])
    m = merge_moments(np.array(recs))
    ref = _records(v)
    np.testing.assert_allclose(m[:3], ref[:3], rtol=1e-12)
    assert m[3] == ref[3]
    assert [shard_size(10, r, 4) for r in range(4)] == [3, 3, 2, 2]


def test_device_side_merge_equals_the_sequential_chan_merge():
    """`merge_moments_t` (torch ops, what the timed job runs on the device) against the NumPy sequential merge: uneven and
    empty shards, several iterations at once, max slots of empty shards ignored."""
    from nf_amd.parallel import merge_moments_t
    rng = np.random.default_rng(3)
    v = rng.normal(2.0, 1e-2, (5, 4000))
    cuts = [0, 0, 1, 1200, 1200, 3999, 4000]
    recs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            x = v[:, a:b]
            m = x.mean(1)
            recs.append(np.stack([np.full(5, b - a, float), m, ((x - m[:, None]) ** 2).sum(1), x.max(1)], -1))
        else:
            recs.append(np.tile(np.array([0.0, 123.0, 0.0, 0.0]), (5, 1)))        # sanitised empty slot with junk in its mean
    R = np.stack(recs, -2)                                                        # (5, shards, 4)
    ref = merge_moments(np.where(R[..., :1] > 0, R, np.array([0.0, 0.0, 0.0, -np.inf])))
    got = merge_moments_t(torch.as_tensor(R)).numpy()
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=1e-12)
    np.testing.assert_array_equal(got[:, 3], v.max(1))
    np.testing.assert_allclose(got[:, 1], v.mean(1), rtol=1e-13)
    np.testing.assert_allclose(got[:, 2] / (4000 - 1), v.var(1, ddof=1), rtol=1e-10)


class _FakeManager:
    """What DataParallel touches of a manager: `_model` and the `data_parallel` slot."""

    def __init__(self, model):
        self._model = model
        self.data_parallel = None


def _dp_worker(rank, ws, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT parameters and buffers
    model = torch.nn.Sequential(torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2)).double()
    with torch.no_grad():
        model[0].running_mean += rank + 1.0
    mgr = _FakeManager(model)
    dp = DataParallel(mgr)
    assert mgr.data_parallel is dp
    start = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [model[0].running_mean.reshape(-1)])
    torch.manual_seed(7 + rank)                         # ... and see different minibatches
    x = torch.rand(64, 3, dtype=torch.double)
    loss = (model(x) ** 2).mean()
    loss.backward()
    local_g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    sc = dp.sync_step(model, torch.stack([loss.detach(), torch.tensor(float(rank), dtype=torch.double)]))
    synced_g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    mx = dp.sync_max(torch.tensor([float(rank), -float(rank)], dtype=torch.double))
    mean = dp.mean(torch.tensor([float(rank)], dtype=torch.double))
    np.savez(os.path.join(outdir, f"dp{rank}.npz"), start=start.numpy(), local_g=local_g.numpy(), synced_g=synced_g.numpy(),
             sc=sc.numpy(), loss=float(loss), mx=mx.numpy(), mean=mean.numpy())
    dist.destroy_process_group()


def test_data_parallel_two_ranks_broadcast_and_one_allreduce_per_step():
    """DataParallel: rank 0's parameters / BN buffers everywhere, gradients and bookkeeping scalars averaged."""
    ws = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_dp_worker, args=(ws, port, td), nprocs=ws, join=True)
        o = [np.load(os.path.join(td, f"dp{r}.npz")) for r in range(ws)]
    np.testing.assert_array_equal(o[0]["start"], o[1]["start"])                 # broadcast from rank 0
    np.testing.assert_array_equal(o[0]["synced_g"], o[1]["synced_g"])           # identical after the all-reduce
    np.testing.assert_allclose(o[0]["synced_g"], (o[0]["local_g"] + o[1]["local_g"]) / 2, rtol=1e-15, atol=1e-18)
    assert np.abs(o[0]["local_g"] - o[1]["local_g"]).max() > 0                  # the shards really differed
    np.testing.assert_allclose(o[0]["sc"], [(o[0]["loss"] + o[1]["loss"]) / 2, 0.5], rtol=1e-15)
    np.testing.assert_array_equal(o[0]["sc"], o[1]["sc"])
    np.testing.assert_array_equal(o[0]["mx"], [1.0, 0.0])
    np.testing.assert_array_equal(o[1]["mean"], [0.5])


def test_data_parallel_single_process_is_a_no_op():
    model = torch.nn.Linear(2, 2).double()
    mgr = _FakeManager(model)
    dp = DataParallel(mgr)
    (model(torch.ones(3, 2, dtype=torch.double)).sum()).backward()
    g = model.weight.grad.clone()
    sc = dp.sync_step(model, torch.tensor([1.5], dtype=torch.double))
    assert torch.equal(model.weight.grad, g) and float(sc[0]) == 1.5
    assert float(dp.sync_max(torch.tensor(2.0))) == 2.0 and float(dp.mean(torch.tensor(2.0))) == 2.0


def test_rccl_init_refuses_more_local_ranks_than_devices(monkeypatch):
    """`init_from_env("nccl")` on a node that shows fewer GPUs than local ranks must fail BEFORE any rendezvous with a message that
    names the fix (two ranks on one device otherwise die deep inside the first collective: `ncclInvalidUsage: Duplicate GPU detected`).
    Runs without a GPU: the device count is what the check reads."""
    from nf_amd import parallel
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.delenv("NIS_DIST_BACKEND", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    called = []
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: called.append((a, k)))
    with pytest.raises(RuntimeError, match="RCCL needs one GPU per rank: 8 local ranks but 1 visible"):
        parallel.init_from_env("nccl")
    assert not called, "no rendezvous may be started"
    # the rehearsal switch takes the same call through gloo, several ranks per device: device = local rank modulo the device count
    monkeypatch.setenv("NIS_DIST_BACKEND", "gloo")
    picked = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: picked.append(d))
    rank, ws, local = parallel.init_from_env("nccl")
    assert (rank, ws, local) == (3, 8, 3) and picked == [0]
    assert len(called) == 1 and called[0][1]["backend"] == "gloo" and "device_id" not in called[0][1]
