"""Where do non-finite / wrong RAMBO outputs sit?  python3 tools/rambo_nan_probe.py [N]   (compares with NIS_RAMBO_REF_LIB if set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gen = FlatInvertiblePhasespace([100.] * 2, [100.] * 4, pdf=None, pdf_active=False)
g = torch.Generator(device=dev).manual_seed(5)
r = torch.rand(N, 8, dtype=torch.double, device=dev, generator=g)
mom = torch.full((N, 6, 4), -7.0, dtype=torch.double, device=dev)
wgt = torch.full((N,), -7.0, dtype=torch.double, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
desc = gen._desc(1000., 0, 0, -1, False)
for rep in range(3):
    gen._launch(desc, r, mom, wgt, st)
torch.cuda.synchronize()
w = wgt.cpu().numpy(); m = mom.cpu().numpy().reshape(N, 24)
badw = np.flatnonzero(~np.isfinite(w)); badm = np.flatnonzero(~np.isfinite(m).all(axis=1))
print("N", N, "status", int(st.item()), "bad weights", badw.size, "bad momenta rows", badm.size, "untouched w", int((w == -7).sum()))
for name, b in (("w", badw), ("m", badm)):
    if b.size:
        print(name, "first", b[:12], "tiles", np.unique(b // 128)[:12], "lanes", np.unique(b % 128)[:16], "n tiles", np.unique(b // 128).size)
# energy-momentum conservation on the good rows
tot = m.reshape(N, 6, 4)[:, 2:, :].sum(axis=1)
ok = np.isfinite(tot).all(axis=1)
print("max |sum p_out - (E,0,0,0)|", np.abs(tot[ok] - np.array([1000., 0, 0, 0])).max(), " <w> good", w[np.isfinite(w)].mean())
if badm.size:
    i = badm[0]; print("row", i, "r =", r[i].cpu().numpy()); print(m[i].reshape(6, 4))
