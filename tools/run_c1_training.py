"""BASELINE config C1 / README example end to end: train 300 epochs (batch = minibatch = 10 000, Adamax, preburn 50),
then integrate.  Reference on 8 CPU threads: 40.3 s, 0.2323 +- 0.0110; integrate(10, 1e5) = 0.2322 +- 4.9e-5."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import integrands
from nisrep.normalizing_flows.manager import *    # the reference's import path

torch.manual_seed(0)
f = integrands.Camel()
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
optim = torch.optim.Adamax(NF._model.parameters(), lr=2e-3, weight_decay=1e-04)
mbs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
with tempfile.TemporaryDirectory() as td:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = NF._train_variance_forward_seq(f, optim, True, td, 10000, 300, 0, False, True, mini_batch_size=mbs,
                                         integrate=True, preburn_time=50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"training: {dt:.2f} s, {len(NF.history)} epochs, returned {res}, best_loss {float(NF.best_loss):.4e} (initial {float(NF.int_loss):.4e})")
t0 = time.perf_counter()
sig, err = NF.integrate(f, 10, 100000, 0)
torch.cuda.synchronize()
print(f"integrate(10, 1e5): {float(sig):.6f} +- {float(err):.2e}  ({time.perf_counter() - t0:.3f} s)   truth 0.2323236")
