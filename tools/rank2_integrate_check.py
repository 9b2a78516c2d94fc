"""Two (or more) ranks of the sample-sharded integrate() job on ONE GPU (NIS_DIST_BACKEND=gloo), running the code an RCCL job runs:
   NIS_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
       tools/rank2_integrate_check.py
* every rank captures its launch set into a hipGraph INSIDE the multi-rank job (IntegratePlan: capture must succeed, not fall back to
  eager launches) and replays it;
* the per-rank (steps, 4) records are combined by `allreduce_moments` on CUDA tensors: ONE gather collective + ONE `nis_merge_moments`
  launch (the branch RCCL takes; gloo serves the gather here) -- checked against a NumPy Chan merge of the same records;
* `integrate_sharded` returns the same numbers on every rank.
Prints "R2_OK ..." on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import integrands, parallel                           # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

rank, ws, local = parallel.init_from_env()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.cuda.set_device(dev)
torch.manual_seed(3)                                 # same parameters on every rank
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3, dev=dev.index)
model = NF.best_model
model.train()
eng = model.engine()
f = integrands.Camel()
G, N = 6, 200_000
plan = eng.plan(G, N, dev, f.nis_kind, f.nis_params)
captured = bool(plan.captured)
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
plan.reset(100 + rank, 0)                            # different draws per rank
plan.run(out, replay=True)
merged = parallel.allreduce_moments(out)             # CUDA tensor: one all-reduce of the zero-padded buffer + nis_merge_moments
# reference: gather the records on the host and merge them with the NumPy Chan update
mine = out.cpu()
allrec = [torch.zeros_like(mine) for _ in range(ws)]
dist.all_gather(allrec, mine)
want = parallel.merge_moments(np.stack([a.numpy() for a in allrec], axis=1))       # (G, world, 4) -> (G, 4)
ok_merge = np.allclose(merged.cpu().numpy(), want, rtol=1e-13, atol=0)
same = [torch.zeros_like(mine) for _ in range(ws)]
dist.all_gather(same, merged.cpu())
ok_same = all(torch.equal(s, same[0]) for s in same)
torch.manual_seed(50 + rank)
sig, err = parallel.integrate_sharded(NF, f, 4, 400_001)
res = torch.tensor([float(sig), float(err)], dtype=torch.double)
allres = [torch.zeros_like(res) for _ in range(ws)]
dist.all_gather(allres, res)
ok_int = all(torch.equal(r, allres[0]) for r in allres) and abs(float(sig) - 0.2323) < 5e-3
ok = captured and ok_merge and ok_same and ok_int
flags = torch.tensor([1 if ok else 0])
dist.all_reduce(flags, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"{'R2_OK' if int(flags) else 'R2_FAIL'} world={ws} backend={dist.get_backend()} captured={captured} merge={ok_merge} "
          f"identical={ok_same} integrate={ok_int} sig={float(sig):.6f}+-{float(err):.2e}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flags) else 1)
