"""Diagnostics-build variants of a C2 launch set against the plain launch-per-phase path of the SAME library:
    NIS_LIB=nf_amd/lib/libnis_hip_diag.so python3 tools/diag_variants_check.py chains2|chains3|resident
Several chains on side streams (NIS_STREAMS=n) and the persistent LDS-resident kernel (NIS_MEGA=1) exist only in
libnis_hip_diag.so (`python -m nf_amd.build --diag`): measured dead ends kept for A/B runs.  Prints OK or raises."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_amd import _lib, integrands                               # noqa: E402
from nf_amd.flow import philox_rows                              # noqa: E402
from nf_amd.normalizing_flows.manager import PWQuadManager       # noqa: E402

variant = sys.argv[1]
assert _lib.lib().nis_diag_build() == 1, "needs the diagnostics library (NIS_LIB=.../libnis_hip_diag.so)"
dev = torch.device("cuda:0")
if variant in ("regbij", "pcwaves"):
    SWITCH = "NIS_REGBIJ" if variant == "regbij" else "NIS_MF_PC"
    # register-resident bijector (flow_mfma_reg.h, NIS_REGBIJ=1) / producer-consumer wave roles (flow_mfma_pc.h, NIS_MF_PC=1) of the
    # matrix-core apply kernel against the product's LDS-image kernel: identical bins, values to rounding (regbij: relative heights,
    # blocked sums; pcwaves: the same arithmetic, bit-identical), for PWQuad K = 16 and PWLin K = 64
    from nf_amd.normalizing_flows.manager import PWLinManager
    for kind in ("pwquad", "pwlin"):
        torch.manual_seed(4)
        if kind == "pwquad":
            NFw = PWQuadManager(n_flow=10)
            NFw.create_model(5, 16, [32] * 3)
        else:
            NFw = PWLinManager(n_flow=8)
            NFw.create_model(4, 4, 64, [32] * 3, 4)
        m = NFw.best_model
        m.train()
        w = torch.rand(20_011, NFw.n_flow, dtype=torch.double, device=dev) * (1 - 1e-9)
        res = {}
        for flag in ("0", "1"):
            os.environ[SWITCH] = flag
            r = m.flow_forward(w, want_xj=True, want_bins=True)
            res[flag] = (r["xj"].cpu().numpy(), r["bins"].cpu().numpy())
        np.testing.assert_array_equal(res["1"][1], res["0"][1])
        np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-11, atol=1e-14)
        f = integrands.Normal(0.5, 0.2, NFw.n_flow)
        outs = {}
        for flag in ("0", "1"):
            os.environ[SWITCH] = flag
            o = torch.zeros((2, 4), dtype=torch.double, device=dev)
            m.engine().integrate_steps_rng(2, 30_001, dev, f.nis_kind, f.nis_params, o, 5, 0)
            outs[flag] = o.cpu().numpy()
        np.testing.assert_allclose(outs["1"], outs["0"], rtol=1e-10)
    print("OK", variant)
    sys.exit(0)
env = {"chains2": {"NIS_STREAMS": "2"}, "chains3": {"NIS_STREAMS": "3"}, "resident": {"NIS_MEGA": "1"}}[variant]
torch.manual_seed(9)
NF = PWQuadManager(n_flow=2)
NF.create_model(2, 4, [3] * 3)
base = NF.best_model
base.train()
f = integrands.Camel()
G, N = 7, 300_017


def run(env, how):
    for k in ("NIS_STREAMS", "NIS_MEGA"):
        os.environ.pop(k, None)
    os.environ.update(env)
    model = copy.deepcopy(base)
    eng = model.engine()
    out = torch.zeros((G, 4), dtype=torch.double, device=dev)
    if how == "fed":
        eng.integrate_steps(philox_rows(77, 3, G, N, 2, dev), True, f.nis_kind, f.nis_params, out)
    else:
        eng.integrate_steps_rng(G, N, dev, f.nis_kind, f.nis_params, out, 77, 3)
    torch.cuda.synchronize()
    eng.resident_status(dev)
    return out.cpu().numpy(), {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


ref, sd_ref = run({}, "rng")
got, sd = run(env, "rng")
np.testing.assert_allclose(got, ref, rtol=1e-12)
for k in sd_ref:
    np.testing.assert_allclose(sd[k], sd_ref[k], rtol=1e-9, atol=1e-15, err_msg=k)
fed, _ = run(env, "fed")
np.testing.assert_allclose(fed, ref, rtol=1e-12)
print("OK", variant)
