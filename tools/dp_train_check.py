"""Data-parallel training rehearsal: launched with
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp_train_check.py
(NIS_DIST_BACKEND=gloo to put both ranks on one GPU).  Every rank trains the README camel model on its own
minibatches; one all-reduce per optimizer step keeps parameters, optimizer state and early-stopping decisions
identical.  Prints "DP_OK ..." on rank 0 when the ranks agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nf_amd import integrands, parallel
from nf_amd.normalizing_flows.manager import PWQuadManager

rank, ws, local = parallel.init_from_env()
dev = torch.device("cuda", local % torch.cuda.device_count())
torch.manual_seed(1000 + rank)                      # different initial parameters AND different minibatches per rank
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3, dev=dev.index)
parallel.DataParallel(NF)
optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-4)
epochs = int(os.environ.get("DP_EPOCHS", "40"))
res = NF._train_variance_forward_seq(integrands.Camel(), optim, False, None, 5000, epochs, 0, True, True, dev=dev.index,
                                     mini_batch_size=5000, integrate=True, preburn_time=10)
flat = torch.cat([p.detach().reshape(-1) for p in NF._model.parameters()] +
                 [b.detach().double().reshape(-1) for b in NF._model.buffers()] +
                 [torch.stack(NF.history).double().reshape(-1), torch.tensor(res, dtype=torch.double, device=dev)])
gathered = [torch.zeros_like(flat) for _ in range(ws)]
dist.all_gather(gathered, flat)
params_equal = all(torch.equal(g[:sum(p.numel() for p in NF._model.parameters())],
                               gathered[0][:sum(p.numel() for p in NF._model.parameters())]) for g in gathered)
hist_equal = all(torch.equal(g[-len(NF.history) - 2:], gathered[0][-len(NF.history) - 2:]) for g in gathered)
# everything, BatchNorm buffers included (running statistics merged whenever a model copy is kept and at the end): model AND best_model
all_equal = all(torch.equal(g, gathered[0]) for g in gathered)
best = torch.cat([t.detach().double().reshape(-1) for t in NF.best_model.state_dict().values()])
gb = [torch.zeros_like(best) for _ in range(ws)]
dist.all_gather(gb, best)
best_equal = all(torch.equal(g, gb[0]) for g in gb)
ok = params_equal and hist_equal and all_equal and best_equal and bool(torch.isfinite(flat).all())
if rank == 0:
    print(f"{'DP_OK' if ok else 'DP_FAIL'} world={ws} epochs_run={len(NF.history)} params_equal={params_equal} buffers_equal={all_equal} best_model_equal={best_equal} "
          f"history_equal={hist_equal} result={res} last_loss={float(NF.history[-1]):.6e}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
