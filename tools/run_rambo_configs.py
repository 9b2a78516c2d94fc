"""RAMBO timings for the three configurations of SURVEY.md 8(d): metric-ii (plain 2->4 massive), C4 (2->4, PDF sampling,
no cuts) and C5 (2->6 massless, PDF sampling, pT / delta-R / eta cuts), N = 1e6, through generateKinematics_batch's launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cases = [("metric-ii plain 2->4", dict(im=[100.] * 2, fm=[100.] * 4, pdf=False), 1000., (0, 0, -1), [0, 0]),
         ("C4 2->4 pdf", dict(im=[0., 0.], fm=[100.] * 4, pdf=True), 13000., (0, 0, -1), [2, -1]),
         ("C5 2->6 pdf+cuts", dict(im=[0., 0.], fm=[0.] * 6, pdf=True), 13000., (20, 0.4, 2.5), [21, 21])]
for name, c, E, cuts, pdgs in cases:
    gen = FlatInvertiblePhasespace(c["im"], c["fm"], pdf=None, pdf_active=c["pdf"])
    r = torch.rand(N, gen.nDimPhaseSpace() + (2 if c["pdf"] else 0), dtype=torch.double, device=dev)
    for _ in range(2):
        mom, wgt = gen.generateKinematics_batch(E, r, cuts[0], cuts[1], cuts[2], pdgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        mom, wgt = gen.generateKinematics_batch(E, r, cuts[0], cuts[1], cuts[2], pdgs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:24s} N={N} {dt * 1e6:8.1f} us/call  {N / dt / 1e9:6.2f} Gsamples/s  pass={float((wgt > 0).double().mean()):.3f} <w>={float(wgt.mean()):.4e}",
          flush=True)
