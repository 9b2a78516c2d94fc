"""Phase timeline of the persistent kernel from a DIAGNOSTIC build (NIS_EXTRA_FLAGS=-DNIS_MEGA_STAMPS python -m nf_amd.build --force):
where a team's batch spends its time (compute phases vs hand-off waits).  Never quote this build's run time.

    python3 tools/mega_stamps.py [N] [G]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import _lib, integrands                              # noqa: E402
from nf_amd import _lib as _nis_lib                               # noqa: E402
assert _nis_lib.lib().nis_diag_build() == 1, "this probe exercises a quarantined variant: run it with NIS_LIB=nf_amd/lib/libnis_hip_diag.so (python -m nf_amd.build --diag)"
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
m = NF.best_model
m.train()
eng = m.engine()
f = integrands.Camel()
out = torch.zeros((G, 4), dtype=torch.double, device=dev)
for _ in range(3):
    eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 1, 0)
torch.cuda.synchronize()
L = _lib.lib()
NT, TEAMS, IL = 256, int(sys.argv[3]) if len(sys.argv) > 3 else 4, 4
n = TEAMS * 256 * IL * 16
buf = (C.c_uint64 * n)()
L.nis_mega_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.nis_mega_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, IL, 16)[:TEAMS * NT].astype(np.float64) / 100.0   # us
names = ["P0 draw+BN0", "handoff0", "stats2", "handoff1", "stats3", "handoff2", "apply0", "handoff3", "stats2'", "handoff4", "stats3'",
         "handoff5", "apply1"]
t0 = st[:, 0, 0].min()
allw = []
for team in range(TEAMS):
    s = st[team * NT:(team + 1) * NT]
    for il in range(min(IL, (G - team + TEAMS - 1) // TEAMS)):
        d = np.diff(s[:, il, :14], axis=1)
        allw.append(d.mean(axis=0))
        print(f"team {team} batch {il}: start {s[:, il, 0].min() - t0:8.1f} .. {s[:, il, 0].max() - t0:8.1f} us, end {s[:, il, 13].min() - t0:8.1f} .. {s[:, il, 13].max() - t0:8.1f}")
        print("    " + "  ".join(f"{nm} {d[:, k].mean():.1f} ({d[:, k].min():.1f}-{d[:, k].max():.1f})" for k, nm in enumerate(names)))
aw = np.array(allw)
print("MEAN over teams/batches: " + "  ".join(f"{nm} {aw[:, k].mean():.1f}" for k, nm in enumerate(names)))
print(f"compute {aw[:, 0::2].sum(axis=1).mean():.1f} us  hand-offs {aw[:, 1::2].sum(axis=1).mean():.1f} us  per batch;  whole launch {st[:, :, 13].max() - t0:.1f} us for {G} batches")
