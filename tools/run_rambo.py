"""Run the 2->4 massive RAMBO kernel a few times (profiling target for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
gen = FlatInvertiblePhasespace([100.] * 2, [100.] * n, pdf=None, pdf_active=False)
gen.bisect_levels = lev
N = 1_000_000
r = torch.rand(N, 3 * n - 4, dtype=torch.double, device=dev)
mom = torch.empty((N, n + 2, 4), dtype=torch.double, device=dev)
wgt = torch.empty(N, dtype=torch.double, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
desc = gen._desc(1000., 0, 0, -1, False)
for _ in range(6):
    gen._launch(desc, r, mom, wgt, st)
torch.cuda.synchronize()
print("ok", float(wgt.mean()))
