"""PyTorch-CPU float64 restatement of the "RAMBO on diet" map (oracle, test / baseline only -- never imported by the product).

The same algorithm as oracle/rambo.py (NumPy), op for op, on torch CPU tensors, so that it runs on ALL host cores through torch's
intra-op threads: the "reference-equivalent CPU path" for part (ii) of the metric (MC samples/s of the 2 -> 4 map) that
bench.py times beside the GPU number.  The reference itself is PyTorch-CPU code with this op mix (one ATen call per arithmetic step, a
lattice bisection of 120-180 levels over the whole batch) and cannot travel to the GPU box.  Pinned to the reference's golden vectors
(tests/golden/rambo.npz): tests/test_oracle_golden.py::test_rambo_torch_restatement_matches_reference.

Reference files restated (relative to /root/reference/nisrep/PhaseSpace):
  flat_phase_space_generator.py:81-97    get_flatWeights              -> flat_weight
  flat_phase_space_generator.py:107-113  rho                          -> rho
  flat_phase_space_generator.py:139-308  generateKinematics_batch     -> generate_kinematics
  flat_phase_space_generator.py:313-359  bisect_vec_batch             -> bisect
  flat_phase_space_generator.py:363-406  intermediate masses + massive reweighting -> inside generate_kinematics
  flat_phase_space_generator.py:408-441  setInitialStateMomenta_batch -> inside generate_kinematics
  utils.py:5-19, 31-36, 58-106, 134-187  set_square, boosts, lab frame, pseudo-rapidity, delta-phi, delta-R
PDFs: pdf=None only (density 1, :124-125), like the bench workloads; the x < 1e-4 cut (:185-186) is kept.
"""
import math

import torch

from . import rambo as _np_rambo

EPS_ANGLE = float(_np_rambo.EPS_ANGLE)
HUGE = float(_np_rambo.HUGE)
ABSOLUTE_ECM_MIN = _np_rambo.ABSOLUTE_ECM_MIN
flat_weight = _np_rambo.flat_weight


def rho(M, N, m):
    """sqrt((M^2-(N+m)^2)(M^2-(N-m)^2)) / (8 M^2)   (:107-113)."""
    Msqr = M ** 2
    return ((Msqr - (N + m) ** 2) * (Msqr - (N - m) ** 2)) ** 0.5 / (8.0 * Msqr)


def bisect(v, n_final, target=1.e-16, max_level=600):
    """v = (e+1) u^e - e u^(e+1), e = n-2..1 per column: lattice bisection in blocks of max_level/10 levels with the batch-global
    stopping rule (:313-359)."""
    if v.shape[1] == 0:
        return None
    e = torch.arange(n_final - 2, 0, step=-1, dtype=torch.double).unsqueeze(0).repeat(v.shape[0], 1)
    level = 0
    left = torch.zeros_like(v)
    right = torch.ones_like(v)
    check = -torch.ones_like(v)
    u = -torch.ones_like(v)
    err = torch.ones_like(v)
    step = max_level / 10
    ml = step
    old = 100
    while float(err.max()) > target and ml < 10 * step:
        while level < ml:
            u = (left + right) * (0.5 ** (level + 1))
            check = (u ** e) * ((e + 1) - e * u)                          # massless_map :101-103
            left = left * 2.0
            right = right * 2.0
            con = torch.full_like(left, 0.5)
            adder = torch.where(v <= check, -con, con)
            left = left + (adder + 0.5)
            right = right + (adder - 0.5)
            level += 1
        err = torch.abs(1.0 - check / v)
        ml = ml + step
        new = float(err.max())
        if new >= old:
            break
        old = new
    return u


def _set_square(p, sq):
    """utils.py:5-19."""
    out = p.clone()
    out[:, 0] = ((p[:, 1:] * p[:, 1:]).sum(-1) + sq) ** 0.5
    return out


def _boost(p, bv):
    """utils.py:58-81 / :83-106 (leg axis broadcast)."""
    b2 = (bv * bv).sum(-1)
    gamma = 1.0 / torch.sqrt(1.0 - b2)
    bp = (p[..., 1:] * bv).sum(-1)
    gamma2 = torch.where(b2 > 0, (gamma - 1.0) / b2, torch.zeros_like(b2))
    factor = gamma2 * bp + gamma * p[..., 0]
    out = p.clone()
    out[..., 1:] = p[..., 1:] + factor.unsqueeze(-1) * bv
    out[..., 0] = gamma * (p[..., 0] + bp)
    return out


def _pseudo_rap(p):
    """utils.py:151-166."""
    pt = torch.sqrt((p[..., 1:3] ** 2).sum(-1))
    th = torch.atan2(pt, p[..., 3])
    eta = -torch.log(torch.tan(th / 2.0))
    return torch.where((pt < EPS_ANGLE) & (torch.abs(p[..., 3]) < EPS_ANGLE), torch.full_like(eta, HUGE), eta)


def _delta_phi(p1, p2):
    """utils.py:170-180."""
    pt1 = torch.sqrt((p1[:, 1:3] ** 2).sum(-1))
    pt2 = torch.sqrt((p2[:, 1:3] ** 2).sum(-1))
    tmp = (p1[:, 1] * p2[:, 1] + p1[:, 2] * p2[:, 2]) / (pt1 * pt2)
    r = torch.where(torch.abs(tmp) > 1.0, torch.acos(tmp / torch.abs(tmp)), torch.acos(torch.clamp(tmp, -1.0, 1.0)))
    return torch.where((pt1 == 0.) | (pt2 == 0.), torch.full_like(r, HUGE), r)


def _delta_R(p1, p2):
    """utils.py:182-187."""
    return torch.sqrt((_pseudo_rap(p1) - _pseudo_rap(p2)) ** 2 + _delta_phi(p1, p2) ** 2)


def generate_kinematics(initial_masses, final_masses, E_cm, r_full, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1,
                        pdf_active=False, tau=True):
    """generateKinematics_batch with pdf=None (:139-308) on torch CPU tensors.  Returns (momenta (N,2+n,4) COM frame, weight (N,))."""
    r_full = torch.as_tensor(r_full, dtype=torch.double)
    if bool(torch.isnan(r_full).any()):
        raise ValueError("NaN random variables")                         # :147-149
    masses = torch.as_tensor(final_masses, dtype=torch.double)
    n = len(final_masses)
    N = r_full.shape[0]
    collider_energy = E_cm
    wgt = torch.ones(N, dtype=torch.double)
    x1 = torch.ones(N, dtype=torch.double)
    x2 = torch.ones(N, dtype=torch.double)
    if not pdf_active:
        r = r_full
        E = E_cm
    else:
        r = r_full[:, :-2]
        if tau:                                                             # :161-176
            tau_min = (max(float(masses.sum()), ABSOLUTE_ECM_MIN) / E_cm) ** 2 * torch.ones(N, dtype=torch.double)
            d_tau = torch.ones(N, dtype=torch.double) - tau_min
            tau_v = tau_min + d_tau * r_full[:, -2]
            y_min = 0.5 * torch.log(tau_v)
            d_y = (-y_min) - y_min
            ycm = y_min + d_y * r_full[:, -1]
            x1 = torch.sqrt(tau_v) * torch.exp(ycm)
            x2 = torch.sqrt(tau_v) * torch.exp(-ycm)
            E = torch.sqrt(tau_v) * E_cm
            wgt = wgt * (d_tau * d_y)
        else:                                                               # :177-182
            x1 = r_full[:, -1].clone()
            x2 = r_full[:, -2].clone()
            E = torch.sqrt(x1 * x2) * E_cm
        x_cut = torch.where(x1 < 1e-4, torch.zeros_like(x1), torch.ones_like(x1))
        x_cut = torch.where(x2 < 1e-4, torch.zeros_like(x_cut), x_cut)
        wgt = wgt * x_cut                                                   # pdf == None -> densities 1  (:187)
    assert r.shape[1] == (0 if n == 1 else 3 * n - 4)                       # :191
    per_sample_E = torch.is_tensor(E)

    # ---- intermediate masses (:379-406, :363-375)
    M = torch.zeros((N, n - 1), dtype=torch.double)
    M[:, 0] = E
    M[:, 0] -= masses.sum()
    u = bisect(r[:, :n - 2], n)
    for i in range(2, n):
        M[:, i - 1] = torch.sqrt(u[:, i - 2] * (M[:, i - 2] ** 2))
    if per_sample_E:
        w_ps = math.pow(2 * math.pi, 4 - 3 * n) * math.pow(math.pi / 2.0, n - 1) * \
            ((E ** 2) ** (n - 2) / (math.factorial(n - 1) * math.factorial(n - 2)))
    else:
        w_ps = torch.full((N,), flat_weight(E, n), dtype=torch.double)
    K = M.clone()
    msum = torch.flip(torch.cumsum(torch.flip(masses, (-1,)), -1), (-1,))
    M = M + msum[:-1]
    w_ps = w_ps * (8. * rho(M[:, n - 2], masses[n - 1], masses[n - 2]))
    w_ps = w_ps * torch.prod((rho(M[:, :n - 2], M[:, 1:], masses[:n - 2]) / rho(K[:, :n - 2], K[:, 1:], 0.)) *
                             (M[:, 1:n - 1] / K[:, 1:n - 1]), -1)
    w_ps = w_ps * torch.pow(K[:, 0] / M[:, 0], 2 * n - 4)
    weight = wgt * w_ps

    # ---- sequential two-body decays (:223-278)
    Q = torch.zeros((N, 4), dtype=torch.double)
    Q[:, 0] = M[:, 0]
    M = torch.cat([M, torch.full((N, 1), float(masses[-1]), dtype=torch.double)], -1)
    q = 4. * M[:, :-1] * rho(M[:, :-1], M[:, 1:], masses[:-1])
    rnd = r[:, n - 2:3 * n - 4]
    cos_t = 2. * rnd[:, 0::2] - 1.
    _theta = torch.acos(cos_t)                                              # computed and unused in the reference as well (:234)
    sin_t = torch.sqrt(1. - cos_t ** 2)
    phi = 2 * math.pi * rnd[:, 1::2]
    cos_p = torch.cos(phi)
    s = torch.sqrt(1. - cos_p ** 2)
    sin_p = torch.where(phi > math.pi, -s, s)
    out = torch.zeros((N, 2 + n, 4), dtype=torch.double)
    for k in range(n - 1):
        p = torch.stack([torch.zeros(N, dtype=torch.double), q[:, k] * sin_t[:, k] * cos_p[:, k], q[:, k] * sin_t[:, k] * sin_p[:, k],
                         q[:, k] * cos_t[:, k]], -1)
        p = _set_square(p, float(masses[k]) ** 2)
        p = _boost(p, Q[:, 1:] / Q[:, :1])
        p = _set_square(p, float(masses[k]) ** 2)
        out[:, 2 + k] = p
        Q = _set_square(Q - p, M[:, k + 1] ** 2)
    out[:, -1] = Q

    # ---- initial state (:408-441)
    if initial_masses[0] == 0. or initial_masses[1] == 0.:
        out[:, 0, 0] = E / 2.0
        out[:, 0, 3] = E / 2.0
        out[:, 1, 0] = E / 2.0
        out[:, 1, 3] = -(E / 2.0)
    else:
        M1, M2 = initial_masses[0] ** 2, initial_masses[1] ** 2
        E1 = (E ** 2 + M1 - M2) / E
        E2 = (E ** 2 - M1 + M2) / E
        Z = (E ** 4 - 2 * E ** 2 * M1 - 2 * E ** 2 * M2 + M1 ** 2 - 2 * M1 * M2 + M2 ** 2) ** 0.5 / E
        out[:, 0, 0] = E1 / 2.0
        out[:, 0, 3] = Z / 2.0
        out[:, 1, 0] = E2 / 2.0
        out[:, 1, 3] = -(Z / 2.0)
    com = out.clone()                                                       # :282

    # ---- lab frame (utils.py:134-146): the whole batch, or none of it when any reference vector has zero spatial part
    ref_lab = out[:, 0, :] * x1.unsqueeze(-1) + out[:, 1, :] * x2.unsqueeze(-1)
    lab = out
    if not bool(((ref_lab[:, 1:] * ref_lab[:, 1:]).sum(-1) == 0).any()):
        lab = _boost(out, (ref_lab[:, 1:] / ref_lab[:, :1]).unsqueeze(1))

    # ---- cuts (:285-301): evaluated whatever their thresholds, like the reference
    pt_min = torch.abs(torch.sqrt(lab[:, 2:, 1] ** 2 + lab[:, 2:, 2] ** 2)).min(dim=1).values
    factor = torch.where(pt_min < pT_mincut, torch.zeros_like(weight), torch.ones_like(weight))
    for i in range(n):
        for j in range(n):
            if i > j:
                factor = factor * torch.where(torch.abs(_delta_R(lab[:, i + 2], lab[:, j + 2])) < delR_mincut,
                                              torch.zeros_like(weight), torch.ones_like(weight))
    if rap_maxcut > 0:
        factor = factor * torch.where(rap_maxcut < torch.abs(_pseudo_rap(lab[:, 2:]).max(dim=1).values),
                                      torch.zeros_like(weight), torch.ones_like(weight))
    weight = weight * factor
    shat = x1 * x2 * collider_energy ** 2
    return com, weight / (2 * shat)
