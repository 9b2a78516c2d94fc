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

from nf_amd.parallel import allreduce_moments, merge_moments, shard_size
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
