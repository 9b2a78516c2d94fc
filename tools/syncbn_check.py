"""Exact-BatchNorm sharding rehearsal (SURVEY.md 8(e) option 1), launched with
   python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P tools/syncbn_check.py
(NIS_DIST_BACKEND=gloo puts the ranks on one GPU).  Every rank forwards ITS rows of one batch with
`FlowEngine.forward_sync`; the concatenated result must equal the single-process train-mode forward of the whole batch
(values, running statistics), which rank 0 computes on a deep copy.  Prints "SYNCBN_OK ..." on rank 0."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from nf_amd import integrands, parallel
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager

rank, ws, local = parallel.init_from_env()
dev = torch.device("cuda", local % torch.cuda.device_count())
ok_all, worst = True, 0.0
cases = [("pwquad", dict(d=2, n_cells=2, K=4, NN=[3, 3, 3])), ("pwquad", dict(d=5, n_cells=3, K=6, NN=[8, 8])),
         ("pwlin", dict(d=6, P=3, n_cells=3, K=20, NN=[7], roll=2)), ("pwquad", dict(d=10, n_cells=5, K=16, NN=[32, 32, 32]))]
for kind, c in cases:
    torch.manual_seed(7)                                   # same model and same batch on every rank
    if kind == "pwquad":
        NF = PWQuadManager(n_flow=c["d"])
        NF.create_model(c["n_cells"], c["K"], list(c["NN"]), dev=dev.index)
    else:
        NF = PWLinManager(n_flow=c["d"])
        NF.create_model(c["P"], c["n_cells"], c["K"], list(c["NN"]), c["roll"])
    with torch.no_grad():
        for p in NF._model.parameters():
            p.add_(0.1 * torch.randn_like(p))
    N = 4099
    w = (torch.rand(N, c["d"], dtype=torch.double, device=dev) * (1 - 1e-9))
    model = NF._model.train()
    ref_model = copy.deepcopy(model).train()
    ref = ref_model.flow_forward(w, want_xj=True)["xj"]          # whole batch, one process
    lo = sum(parallel.shard_size(N, r, ws) for r in range(rank))
    hi = lo + parallel.shard_size(N, rank, ws)
    mine = model.engine().forward_sync(w[lo:hi].contiguous(), want_xj=True)["xj"]
    err = float((mine - ref[lo:hi]).abs().max() / ref.abs().max())
    sd, sdr = model.state_dict(), ref_model.state_dict()
    err_rs = max(float((sd[k] - sdr[k]).abs().max() / (sdr[k].abs().max() + 1e-300)) for k in sd if "running" in k)
    trk = all(int(sd[k]) == int(sdr[k]) for k in sd if "num_batches" in k)
    t = torch.tensor([err, err_rs, 0.0 if trk else 1.0], dtype=torch.double, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = float(t[0]) < 1e-11 and float(t[1]) < 1e-11 and float(t[2]) == 0.0
    ok_all &= ok
    worst = max(worst, float(t[0]), float(t[1]))
    if rank == 0:
        print(f"{kind} d={c['d']}: max |dXJ| rel {float(t[0]):.2e}, running stats rel {float(t[1]):.2e}, tracked ok {trk} -> {'ok' if ok else 'FAIL'}",
              flush=True)
# sharded integrate with exact statistics == the same estimator structure (finite, same on all ranks)
torch.manual_seed(11)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3, dev=dev.index)
sig, err = parallel.integrate_sharded(NF, integrands.Camel(), 3, 20001, sync_bn=True)
g = torch.tensor([float(sig), float(err)], dtype=torch.double, device=dev)
gl = [torch.zeros_like(g) for _ in range(ws)]
dist.all_gather(gl, g)
same = all(torch.equal(x, gl[0]) for x in gl)
ok_all &= same and bool(torch.isfinite(g).all())
if rank == 0:
    print(f"{'SYNCBN_OK' if ok_all else 'SYNCBN_FAIL'} world={ws} worst={worst:.2e} integrate={float(sig):.6f}+-{float(err):.1e} same_on_all_ranks={same}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok_all else 1)
