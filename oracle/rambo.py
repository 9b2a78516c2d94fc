"""NumPy float64 restatement of the "RAMBO on diet" map of NGoetz/NF (oracle, test-only).

Reference files restated (relative to /root/reference/nisrep/PhaseSpace):
  flat_phase_space_generator.py:81-97    get_flatWeights            -> flat_weight
  flat_phase_space_generator.py:107-113  rho                        -> rho
  flat_phase_space_generator.py:139-308  generateKinematics_batch   -> generate_kinematics
  flat_phase_space_generator.py:313-359  bisect_vec_batch           -> bisect  (lattice bisection, batch-global stop)
  flat_phase_space_generator.py:363-406  intermediates massless/massive -> inside generate_kinematics
  flat_phase_space_generator.py:408-441  setInitialStateMomenta_batch
  utils.py:5-19,31-36,58-106,124-187     set_square, boosts, uniform_distr, pseudo-rapidity, delta-phi, delta-R

PDFs: only pdf=None is restated (density == 1, flat_phase_space_generator.py:124-125); the
x < 1e-4 cut (:185-186) is kept.
"""
import math
import numpy as np

EPS_ANGLE = np.finfo(float).eps ** 0.5      # utils.py:151,160,170 default eps
HUGE = np.finfo(float).max                   # utils.py default huge
ABSOLUTE_ECM_MIN = 1.0                       # flat_phase_space_generator.py:68


def flat_weight(E_cm, n):
    """(2pi)^(4-3n) (pi/2)^(n-1) (E^2)^(n-2) / ((n-1)!(n-2)!)   (:81-97)."""
    if n == 1:
        return 1.0
    return math.pow(2 * math.pi, 4 - 3 * n) * math.pow(math.pi / 2.0, n - 1) * \
        ((E_cm ** 2) ** (n - 2) / (math.factorial(n - 1) * math.factorial(n - 2)))


def rho(M, N, m):
    """sqrt((M^2-(N+m)^2)(M^2-(N-m)^2)) / (8 M^2)   (:107-113)."""
    Msqr = M ** 2
    return ((Msqr - (N + m) ** 2) * (Msqr - (N - m) ** 2)) ** 0.5 / (8.0 * Msqr)


def bisect(v, n_final, target=1.e-16, max_level=600, return_levels=False):
    """Solve v = (e+1) u^e - e u^(e+1), e = n-2..1 per column, by bisection on an integer lattice
    in blocks of max_level/10 levels; stop when the BATCH maximum relative error is <= target or
    stops improving (:313-359)."""
    if v.shape[1] == 0:
        return (None, 0) if return_levels else None
    e = np.arange(n_final - 2, 0, -1, dtype=np.float64)[None, :].repeat(v.shape[0], 0)
    level = 0
    left = np.zeros_like(v)
    right = np.ones_like(v)
    check = -np.ones_like(v)
    u = -np.ones_like(v)
    err = np.ones_like(v)
    step = max_level / 10
    ml = step
    old = 100
    while err.max() > target and ml < 10 * step:
        while level < ml:
            u = (left + right) * (0.5 ** (level + 1))
            check = (u ** e) * ((e + 1) - e * u)                    # massless_map :101-103
            left = left * 2.0
            right = right * 2.0
            adder = np.where(v <= check, -0.5, 0.5)
            left = left + (adder + 0.5)
            right = right + (adder - 0.5)
            level += 1
        with np.errstate(divide="ignore", invalid="ignore"):
            err = np.abs(1.0 - check / v)
        ml = ml + step
        new = err.max()
        if new >= old:
            break
        old = new
    return (u, level) if return_levels else u


def _set_square(p, sq):
    """utils.py:5-19: energy := sqrt(|p|^2 + square)."""
    out = p.copy()
    out[:, 0] = ((p[:, 1:] * p[:, 1:]).sum(-1) + sq) ** 0.5
    return out


def _boost(p, bv):
    """utils.py:58-81 (and :83-106 with a broadcast leg axis): boost p by velocity bv."""
    b2 = (bv * bv).sum(-1)
    gamma = 1.0 / np.sqrt(1.0 - b2)
    bp = (p[..., 1:] * bv).sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        gamma2 = np.where(b2 > 0, (gamma - 1.0) / b2, 0.0)
    factor = gamma2 * bp + gamma * p[..., 0]
    out = p.copy()
    out[..., 1:] = p[..., 1:] + factor[..., None] * bv
    out[..., 0] = gamma * (p[..., 0] + bp)
    return out


def _pseudo_rap(p):
    """utils.py:151-166."""
    pt = np.sqrt((p[..., 1:3] ** 2).sum(-1))
    th = np.arctan2(pt, p[..., 3])
    with np.errstate(divide="ignore", invalid="ignore"):
        eta = -np.log(np.tan(th / 2.0))
    return np.where((pt < EPS_ANGLE) & (np.abs(p[..., 3]) < EPS_ANGLE), HUGE, eta)


def _delta_phi(p1, p2):
    """utils.py:170-180."""
    pt1 = np.sqrt((p1[:, 1:3] ** 2).sum(-1))
    pt2 = np.sqrt((p2[:, 1:3] ** 2).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        tmp = (p1[:, 1] * p2[:, 1] + p1[:, 2] * p2[:, 2]) / (pt1 * pt2)
        r = np.where(np.abs(tmp) > 1.0, np.arccos(tmp / np.abs(tmp)), np.arccos(np.clip(tmp, -1.0, 1.0)))
    return np.where((pt1 == 0.) | (pt2 == 0.), HUGE, r)


def _delta_R(p1, p2):
    """utils.py:182-187."""
    with np.errstate(over="ignore", invalid="ignore"):
        return np.sqrt((_pseudo_rap(p1) - _pseudo_rap(p2)) ** 2 + _delta_phi(p1, p2) ** 2)


def generate_kinematics(initial_masses, final_masses, E_cm, r_full, pT_mincut=-1, delR_mincut=-1, rap_maxcut=-1,
                        pdf_active=False, tau=True, want=None, pdf=None, pdgs=(0, 0)):
    """generateKinematics_batch with pdf=None (:139-308).  Returns (momenta (N,2+n,4) in the COM frame,
    weight (N,)).  `want` (dict) receives intermediates: u, M (with masses), x1, x2, cut factor, lab momenta."""
    r_full = np.asarray(r_full, dtype=np.float64)
    if np.isnan(r_full).any():
        raise ValueError("NaN random variables")                         # :147-149
    masses = np.asarray(final_masses, dtype=np.float64)
    n = len(final_masses)
    N = r_full.shape[0]
    collider_energy = E_cm
    wgt = np.ones(N)
    x1 = np.ones(N)
    x2 = np.ones(N)
    if not pdf_active:
        r = r_full
        E = E_cm                                                            # python float
    else:
        r = r_full[:, :-2]
        if tau:                                                             # :161-176
            tau_min = (max(float(masses.sum()), ABSOLUTE_ECM_MIN) / E_cm) ** 2 * np.ones(N)
            d_tau = np.ones(N) - tau_min
            tau_v = tau_min + d_tau * r_full[:, -2]
            y_min = 0.5 * np.log(tau_v)
            d_y = (-y_min) - y_min
            ycm = y_min + d_y * r_full[:, -1]
            x1 = np.sqrt(tau_v) * np.exp(ycm)
            x2 = np.sqrt(tau_v) * np.exp(-ycm)
            E = np.sqrt(tau_v) * E_cm
            wgt = wgt * (d_tau * d_y)
        else:                                                               # :177-182
            x1 = r_full[:, -1].copy()
            x2 = r_full[:, -2].copy()
            E = np.sqrt(x1 * x2) * E_cm
        x_cut = np.where(x1 < 1e-4, 0.0, 1.0)
        x_cut = np.where(x2 < 1e-4, 0.0, x_cut)
        wgt = wgt * x_cut                                                   # pdf == None -> densities 1  (:187)
        if pdf is not None:                                                 # get_pdfQ2 (:120-137) at the fixed scale 91.188^2 (:184);
            from . import pdf as _pdf                                       # `pdf`: list of oracle.pdf.Subgrid (parity unpinned, see there)
            wgt = wgt * _pdf.density(pdf, pdgs[0], x1, 91.188 ** 2) * _pdf.density(pdf, pdgs[1], x2, 91.188 ** 2)
    assert r.shape[1] == (0 if n == 1 else 3 * n - 4)                       # :191
    per_sample_E = isinstance(E, np.ndarray)

    # ---- intermediate masses (:379-406, :363-375)
    M = np.zeros((N, n - 1))
    M[:, 0] = E
    M[:, 0] -= masses.sum()
    u = bisect(r[:, :n - 2], n)
    for i in range(2, n):
        M[:, i - 1] = np.sqrt(u[:, i - 2] * (M[:, i - 2] ** 2))
    if per_sample_E:
        w_ps = math.pow(2 * math.pi, 4 - 3 * n) * math.pow(math.pi / 2.0, n - 1) * \
            ((E ** 2) ** (n - 2) / (math.factorial(n - 1) * math.factorial(n - 2)))
    else:
        w_ps = np.full(N, flat_weight(E, n))
    K = M.copy()
    msum = np.cumsum(masses[::-1])[::-1]
    M = M + msum[:-1]
    w_ps = w_ps * (8. * rho(M[:, n - 2], masses[n - 1], masses[n - 2]))
    with np.errstate(divide="ignore", invalid="ignore"):
        w_ps = w_ps * np.prod((rho(M[:, :n - 2], M[:, 1:], masses[:n - 2]) / rho(K[:, :n - 2], K[:, 1:], 0.)) *
                              (M[:, 1:n - 1] / K[:, 1:n - 1]), -1)
        w_ps = w_ps * (K[:, 0] / M[:, 0]) ** (2 * n - 4)
    weight = wgt * w_ps

    # ---- sequential two-body decays (:223-278)
    Q = np.zeros((N, 4))
    Q[:, 0] = M[:, 0]
    M = np.concatenate([M, np.full((N, 1), masses[-1])], -1)
    with np.errstate(invalid="ignore"):
        q = 4. * M[:, :-1] * rho(M[:, :-1], M[:, 1:], masses[:-1])
    rnd = r[:, n - 2:3 * n - 4]
    cos_t = 2. * rnd[:, 0::2] - 1.
    sin_t = np.sqrt(1. - cos_t ** 2)
    phi = 2 * math.pi * rnd[:, 1::2]
    cos_p = np.cos(phi)
    s = np.sqrt(1. - cos_p ** 2)
    sin_p = np.where(phi > math.pi, -s, s)
    out = np.zeros((N, 2 + n, 4))
    for k in range(n - 1):
        p = np.stack([np.zeros(N), q[:, k] * sin_t[:, k] * cos_p[:, k], q[:, k] * sin_t[:, k] * sin_p[:, k],
                      q[:, k] * cos_t[:, k]], -1)
        p = _set_square(p, masses[k] ** 2)
        p = _boost(p, Q[:, 1:] / Q[:, :1])
        p = _set_square(p, masses[k] ** 2)
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
        Z = np.sqrt(E ** 4 - 2 * E ** 2 * M1 - 2 * E ** 2 * M2 + M1 ** 2 - 2 * M1 * M2 + M2 ** 2) / E
        out[:, 0, 0] = E1 / 2.0
        out[:, 0, 3] = Z / 2.0
        out[:, 1, 0] = E2 / 2.0
        out[:, 1, 3] = -(Z / 2.0)
    com = out.copy()                                                        # :282

    # ---- boost to the lab frame (utils.py:134-146): applied to the whole batch, or to none of it
    # when any sample has a reference vector with zero spatial part (always so without PDFs).
    ref_lab = out[:, 0, :] * x1[:, None] + out[:, 1, :] * x2[:, None]
    lab = out
    if not ((ref_lab[:, 1:] * ref_lab[:, 1:]).sum(-1) == 0).any():
        lab = _boost(out, (ref_lab[:, 1:] / ref_lab[:, :1])[:, None, :])

    # ---- cuts (:285-301)
    pt_min = np.abs(np.sqrt(lab[:, 2:, 1] ** 2 + lab[:, 2:, 2] ** 2)).min(axis=1)
    factor = np.where(pt_min < pT_mincut, 0.0, 1.0)
    for i in range(n):
        for j in range(n):
            if i > j:
                factor = factor * np.where(np.abs(_delta_R(lab[:, i + 2], lab[:, j + 2])) < delR_mincut, 0.0, 1.0)
    if rap_maxcut > 0:
        factor = factor * np.where(rap_maxcut < np.abs(_pseudo_rap(lab[:, 2:]).max(axis=1)), 0.0, 1.0)
    weight = weight * factor
    shat = x1 * x2 * collider_energy ** 2
    if want is not None:
        want.update(u=u, M=M, x1=x1, x2=x2, factor=factor, lab=lab)
    return com, weight / (2 * shat)
