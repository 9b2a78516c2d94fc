"""`FlatInvertiblePhasespace` with the reference's constructor and `generateKinematics_batch` signature
(/root/reference/nisrep/PhaseSpace/flat_phase_space_generator.py:25-54, :57-79, :139-308); the map itself
runs in the `k_rambo` HIP kernel (nis_rambo_forward).  pdf=None gives unit parton densities (reference :124-125);
`pdf=nf_amd.PhaseSpace.pdf_grid.PdfGrid` (an lhagrid1 grid, what LHAPDF reads) evaluates x f(x, 91.188^2) / x per beam
inside the kernel (reference :120-137, :183-187; LHAPDF itself stays out of scope).
"""
import ctypes as C
import math

import torch

from .. import _lib


class PhaseSpaceGeneratorError(Exception):
    pass


class VirtualPhaseSpaceGenerator(object):
    def __init__(self, initial_masses, final_masses, pdf=None, pdf_active=False, tau=True):
        self.initial_masses = initial_masses
        self.masses_t = torch.tensor(final_masses, requires_grad=False, dtype=torch.double)
        self.n_initial = len(initial_masses)
        self.n_final = len(final_masses)
        self.pdf = pdf
        self.pdf_active = pdf_active
        self.tau = tau
        if pdf is not None and not (hasattr(pdf, "table") and hasattr(pdf, "has")):
            raise NotImplementedError("pdf must be None or an nf_amd.PhaseSpace.pdf_grid.PdfGrid (an lhagrid1 grid evaluated on "
                                      "the device); lhapdf.PDF objects are host-only and out of scope")

    def generateKinematics(self, E_cm, random_variables):
        raise NotImplementedError

    def nDimPhaseSpace(self):
        """Number of random numbers a point needs (reference :48-54)."""
        if self.n_final == 1:
            return 0
        return 3 * self.n_final - 4


class FlatInvertiblePhasespace(VirtualPhaseSpaceGenerator):
    """RAMBO on diet, S. Plaetzer arXiv:1308.2922."""

    epsilon_border = 1e-10
    absolute_Ecm_min = 1.
    bisect_levels = 0          # 0: monotone Newton solver (default); > 0: the reference's lattice bisection with that many
                               # levels per sample (the reference executes a batch-dependent 120-180)

    def __init__(self, *args, **opts):
        super().__init__(*args, **opts)
        if self.n_initial == 1:
            raise PhaseSpaceGeneratorError("This basic generator does not support decay topologies.")
        if self.n_initial > 2:
            raise PhaseSpaceGeneratorError("This basic generator does not support more than 2 incoming particles.")
        if self.n_final < 2 or self.n_final > _lib.NIS_MAX_LEGS:
            raise PhaseSpaceGeneratorError(f"nf_amd supports 2..{_lib.NIS_MAX_LEGS} final-state particles")

    @staticmethod
    def get_flatWeights(E_cm, n):
        """(2pi)^(4-3n) (pi/2)^(n-1) (E_cm^2)^(n-2) / ((n-1)!(n-2)!)   (reference :81-97)."""
        if n == 1:
            return 1.
        pref = math.pow(2 * math.pi, 4 - 3 * n) * math.pow((math.pi / 2.0), n - 1)
        fact = math.factorial(n - 1) * math.factorial(n - 2)
        if not torch.is_tensor(E_cm):
            return pref * (math.pow((E_cm ** 2), n - 2) / fact)
        return pref * (torch.pow((E_cm ** 2), n - 2) / fact)

    def _desc(self, E_cm, pT_mincut, delR_mincut, rap_maxcut, lab_boost):
        d = _lib.RamboDesc()
        d.n_final = self.n_final
        d.flags = (_lib.NIS_RAMBO_PDF_ACTIVE if self.pdf_active else 0) | (_lib.NIS_RAMBO_TAU if self.tau else 0) | \
                  (_lib.NIS_RAMBO_LAB_BOOST if lab_boost else 0)
        d.initial_masses[0], d.initial_masses[1] = float(self.initial_masses[0]), float(self.initial_masses[1])
        for i, m in enumerate(self.masses_t.tolist()):
            d.final_masses[i] = float(m)
        d.E_cm = float(E_cm)
        d.pT_mincut, d.delR_mincut, d.rap_maxcut = float(pT_mincut), float(delR_mincut), float(rap_maxcut)
        d.bisect_levels = int(self.bisect_levels)
        return d

    PDF_SCALE2 = 91.188 ** 2   # the reference evaluates the densities at this fixed scale (:184)

    def _pdf_tables(self, pdgs, dev):
        """nis_pdf1d[2] for the two beams, or None: get_pdfQ2 (:120-137) uses the grid for the gluon and the quarks only."""
        if self.pdf is None or not self.pdf_active:
            return None
        from .pdf_grid import Pdf1d
        tabs = (Pdf1d * 2)()
        for b in range(2):
            pdg = int(pdgs[b])
            if pdg == 21 or abs(pdg) in range(1, 7):
                if not self.pdf.has(pdg):
                    raise PhaseSpaceGeneratorError(f"the PDF grid holds no parton {pdg}")
                tabs[b] = self.pdf.table(pdg, self.PDF_SCALE2, dev)
        return tabs

    def _launch(self, desc, r, momenta, weight, status, col_major=False, cols=None, ld=None, N=None, pdgs=(0, 0)):
        L = _lib.lib()
        N = r.shape[0] if N is None else N
        ld = r.shape[1] if ld is None else ld
        cp = (C.c_int32 * len(cols))(*cols) if cols is not None else None
        tabs = self._pdf_tables(pdgs, r.device)
        with _lib.on_device(r.device):
            rc = L.nis_rambo_forward_pdf(C.byref(desc), tabs, r.data_ptr(), ld, 1 if col_major else 0, cp, N,
                                         momenta.data_ptr() if momenta is not None else None, weight.data_ptr(),
                                         status.data_ptr(), _lib.stream_ptr(r.device))
        _lib.check(rc, "nis_rambo_forward_pdf")

    def generateKinematics_batch(self, E_cm, random_variables_full, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1,
                                 pdgs=[0, 0]):
        """(N, 3n-4 [+2]) uniform numbers -> (momenta (N, 2+n, 4) in the COM frame, weight (N,)), float64, on the
        input's device (which must be the GPU).  Cuts act on the lab-frame copy (reference :282-308)."""
        _lib.require_gpu()
        r = random_variables_full
        if not torch.is_tensor(r) or not r.is_cuda:
            raise _lib.NisError("generateKinematics_batch needs a CUDA tensor of random variables (no CPU fallback)")
        r = r.detach().to(torch.double).contiguous()
        ncol = self.nDimPhaseSpace() + (2 if self.pdf_active else 0)
        assert r.dim() == 2 and r.shape[1] == ncol
        self.masses_t = self.masses_t.to(r.device)
        self.collider_energy = E_cm
        N = r.shape[0]
        momenta = torch.empty((N, self.n_initial + self.n_final, 4), dtype=torch.double, device=r.device)
        weight = torch.empty(N, dtype=torch.double, device=r.device)
        status = torch.zeros(1, dtype=torch.int32, device=r.device)
        # Lab boost: the reference boosts the WHOLE batch unless some sample has a zero spatial reference
        # vector (utils.py:139-144) -- always the case without PDFs (x1 = x2 = 1).
        lab = bool(self.pdf_active)
        self._launch(self._desc(E_cm, pT_mincut, delR_mincut, rap_maxcut, lab), r, momenta, weight, status, pdgs=pdgs)
        st = int(status.item())
        if st & 2:
            raise PhaseSpaceGeneratorError("Some of the random variables passed to the phase-space generator are NaN")
        if lab and (st & 4):          # a sample with x1 == x2: the reference then skips the boost for everyone
            status.zero_()
            self._launch(self._desc(E_cm, pT_mincut, delR_mincut, rap_maxcut, False), r, momenta, weight, status, pdgs=pdgs)
            st = int(status.item())
        if st & 8:
            print("Invalid boost")    # utils.py:33-34
        return momenta, weight

    def weights_from_planes(self, E_cm, planes, cols, N, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1, pdgs=(0, 0)):
        """Weight only, reading the random numbers from column planes (the flow's internal layout):
        the integrand side of a fused flow -> RAMBO evaluation.  No momenta are written."""
        _lib.require_gpu()
        ncol = self.nDimPhaseSpace() + (2 if self.pdf_active else 0)
        assert len(cols) == ncol, f"the flow has {len(cols)} columns, the phase space needs {ncol}"
        weight = torch.empty(N, dtype=torch.double, device=planes.device)
        status = torch.zeros(1, dtype=torch.int32, device=planes.device)
        self.collider_energy = E_cm
        lab = bool(self.pdf_active)
        desc = self._desc(E_cm, pT_mincut, delR_mincut, rap_maxcut, lab)
        self._launch(desc, planes, None, weight, status, col_major=True, cols=list(cols), ld=planes.shape[1], N=N, pdgs=pdgs)
        # the status word is honoured exactly as on the row path (generateKinematics_batch): NaN guard (reference :147-149),
        # and the batch-global "skip the lab boost if any sample has x1 == x2" rule (utils.py:139-144)
        st = int(status.item())
        if st & 2:
            raise PhaseSpaceGeneratorError("Some of the random variables passed to the phase-space generator are NaN")
        if lab and (st & 4):
            status.zero_()
            self._launch(self._desc(E_cm, pT_mincut, delR_mincut, rap_maxcut, False), planes, None, weight, status, col_major=True,
                         cols=list(cols), ld=planes.shape[1], N=N, pdgs=pdgs)
            st = int(status.item())
        if st & 8:
            print("Invalid boost")    # utils.py:33-34
        return weight, status

    def moments_from_planes(self, E_cm, planes, cols, jplane, N, out4, status, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1, pdgs=(0, 0),
                            lab_boost=None):
        """weight * J -> (count, mean, M2, max) for G plane sets in ONE launch + one merge launch (`nis_rambo_moments`): the
        integrand side of integrate()'s loop body with f = this generator's weight (manager.py:397-400).  planes: (G, d+1, N)
        as `FlowEngine.integrate_steps*` leaves them with keep_planes; out4: (G, 4) device view; status: device int32[1] that
        ACCUMULATES the kernel's status bits -- nothing is read back here (no host sync, no allocation besides the cached
        workspace); the caller inspects `status` once per integrate() call (`check_status`)."""
        _lib.require_gpu()
        ncol = self.nDimPhaseSpace() + (2 if self.pdf_active else 0)
        assert len(cols) == ncol, f"the flow has {len(cols)} columns, the phase space needs {ncol}"
        L = _lib.lib()
        G = planes.shape[0]
        dev = planes.device
        wsb = L.nis_rambo_moments_workspace_bytes(G)
        ws = getattr(self, "_mom_ws", None)
        if ws is None or ws.device != dev or ws.numel() < wsb:
            self._mom_ws = ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        self.collider_energy = E_cm
        lab = bool(self.pdf_active) if lab_boost is None else bool(lab_boost)
        desc = self._desc(E_cm, pT_mincut, delR_mincut, rap_maxcut, lab)
        cp = (C.c_int32 * len(cols))(*cols)
        tabs = self._pdf_tables(pdgs, dev)
        with _lib.on_device(dev):
            rc = L.nis_rambo_moments(C.byref(desc), tabs, planes.data_ptr(), planes.shape[2], planes.stride(0), cp, int(jplane), int(N), G,
                                     None, out4.data_ptr(), status.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "nis_rambo_moments")

    def check_status(self, status):
        """Honour the accumulated status word exactly as `generateKinematics_batch` does per call: NaN guard (reference :147-149)
        raises; returns True when some sample had x1 == x2 under the lab boost -- the reference then skips the boost for that
        whole batch (utils.py:139-144), so the caller has to redo its estimate batch by batch."""
        st = int(status.item())
        if st & 2:
            raise PhaseSpaceGeneratorError("Some of the random variables passed to the phase-space generator are NaN")
        if st & 8:
            print("Invalid boost")    # utils.py:33-34
        return bool(self.pdf_active) and bool(st & 4)
