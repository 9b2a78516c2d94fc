"""Fixed vs per-sample cost of the C2 kernels: per-class HIP-event time of integrate_steps for several batch sizes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd import _lib, integrands
from nf_amd.normalizing_flows.manager import PWQuadManager
dev = torch.device("cuda:0")
torch.manual_seed(0)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
eng = NF.best_model.train().engine()
f = integrands.Camel()
L = _lib.lib()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = {"apply": 0, "stats": 1, "ingest": 2}
NS = [int(float(v)) for v in os.environ.get("SCAN_N", "1e3,1e4,1e5,3e5,1e6,3e6").split(",")]
for N in NS:
    w = torch.rand(G, N, 2, dtype=torch.float32, device=dev)
    out = torch.zeros(G, 4, dtype=torch.double, device=dev)
    tok = eng.pack_token()
    for _ in range(3):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
    line = [f"N={N:8d} G={G}"]
    for name, cls in K.items():
        _lib.check(L.nis_prof_enable(cls, 256), "prof")
        for _ in range(5):
            eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
        torch.cuda.synchronize()
        ms, n = C.c_double(0.0), C.c_int32(0)
        L.nis_prof_read(C.byref(ms), C.byref(n))
        L.nis_prof_disable()
        line.append(f"{name} {ms.value / max(n.value, 1) * 1e3:7.1f} us x{n.value // 5}/call")
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        eng.integrate_steps(w, True, f.nis_kind, f.nis_params, out, token=tok)
    torch.cuda.synchronize()
    line.append(f"call {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us")
    print("  ".join(line), flush=True)
