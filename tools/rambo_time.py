"""2->4 massive RAMBO kernel time (events around 200 launches on 4 rotating buffer sets).  python3 tools/rambo_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
dev = torch.device("cuda:0")
N = 1_000_000
gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
r = [torch.rand(N, 8, dtype=torch.double, device=dev) for _ in range(4)]
mom = [torch.empty((N, 6, 4), dtype=torch.double, device=dev) for _ in range(4)]
wgt = [torch.empty(N, dtype=torch.double, device=dev) for _ in range(4)]
st = torch.zeros(1, dtype=torch.int32, device=dev)
desc = gen._desc(1000., 0, 0, -1, False)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(50):
        gen._launch(desc, r[k % 4], mom[k % 4], wgt[k % 4], st)
    e0.record()
    for k in range(200):
        gen._launch(desc, r[k % 4], mom[k % 4], wgt[k % 4], st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
print(f"{os.environ.get('NIS_LIB', 'default').split('/')[-1]:16s} {us:7.2f} us/launch  {264e6 / us / 1e6:6.2f} TB/s  <w>={float(wgt[0].mean()):.5f}")
