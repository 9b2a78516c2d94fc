"""Phase timeline of the matrix-core apply kernel from a STAMPED variant build:
   tools/build_variant.sh stamps "-DNIS_WM=32 -DNIS_MF_STAMPS" flow_wm flow_wm32 ; NIS_LIB=$PWD/nf_amd/lib/variants/stamps.so python3 tools/wide_stamps.py C5
Workgroup 1's four waves report shader cycles spent in: hidden stack (input + 3 GEMMs + BN/ReLU) | last-layer tiles (MFMA chain + exp +
image writes) | bijector | tail (Jacobian product, stores, integrand rider), and the wave's lifetime, for the LAST apply launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
kind, d, kw, N, _, _ = bench.CONFIGS[name]
torch.manual_seed(10)
NF = (bench.PWQuadManager if kind == "pwquad" else bench.PWLinManager)(n_flow=d)
NF.create_model(**kw)
m = NF.best_model
m.train()
eng = m.engine()
f = bench.integrands.Normal(0.5, 0.1, d)
out = torch.zeros((1, 4), dtype=torch.double, device=dev)
for k in range(2):
    eng.integrate_steps_rng(1, N, dev, f.nis_kind, f.nis_params, out, 1, k)
torch.cuda.synchronize()
st = eng._ws[64:64 + 4 * 5 * 8].view(torch.int64).cpu().numpy().reshape(4, 5)
tiles = N / (512 * 4 * 16 * 2)        # sample-tile pairs per wave (2 workgroups per CU)
for w in range(4):
    h, t, b, tl, life = st[w]
    print(f"[{name}] wave {w}: hidden {h:>10d}  tiles {t:>10d}  bijector {b:>10d}  tail {tl:>10d}  | sum {h+t+b+tl:>10d} of lifetime {life:>10d}"
          f"  ({100*h/life:.0f} / {100*t/life:.0f} / {100*b/life:.0f} / {100*tl/life:.0f} %)")
