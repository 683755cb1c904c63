"""CPU oracle for the SPH inner loop of dmuley/sph-code  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The product path
(`sph_code_amd`) never imports anything from `oracle/` and fails loudly when the
HIP library is missing.

It is an independent NumPy restatement (written from the formulas in SURVEY.md
Appendix A, not transcribed from the source) of the reference's hot path.
"nsc" = /root/reference/sph/navier_stokes_cleaned.py, "drv" =
/root/reference/sph/code_running.py.

Parity pinning: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, run in the build
container by `tests/golden/make_golden.py` and committed as `tests/golden/*.npz`
(`tests/test_oracle_golden.py`).  The integrator and time-step control (`timestep`,
`clamp_state`, `assemble_loop`, `leapfrog`; drv:222-238,460-491) are pinned the same way:
the driver script as a whole is not runnable, so `tests/golden/make_golden_driver.py`
executes those two statement blocks of it, taken by line range, on seeded arrays
(`tests/golden/driver_integrator.npz`).  One part stays "parity unpinned" by the
reference: the choice of which neighbours an eps=0.1 cKDTree traversal returns (SciPy-internal).  `gravity_direct` is
the softened direct sum the reference's tree gravity (nsc:252-415) approximates: that
routine can neither be run nor pinned, so parity with IT is unpinned and the sum is
checked against closed forms only.  `clip_grad=True` in `hydro_update` / `step` is a
physics option of the build, not the reference's arithmetic.
"""
import numpy as np

# --- constants read by the path (nsc:21-52); numeric values as used for the fixtures ------
K_B = 1.380649e-23                 # scipy.constants.Boltzmann            nsc:22
AMU = 1.66053906892e-27            # 'atomic mass constant' (CODATA 2022) nsc:25
M_H = 1.0008 * AMU                 # nsc:29
SOLAR_MASS = 1.989e30              # nsc:30
M_0 = 10 ** 1.5 * SOLAR_MASS       # nsc:36
AU = 149597870700.0                # nsc:24
YEAR = 60. * 60. * 24. * 365.      # nsc:37
DT_0 = YEAR * 250000.              # nsc:38
MU_SPECIE = np.array([2.0158, 4.0026, 1.0079, 1.0074, 4.0021, 4.0016, 0.0005, 140.69, 60.08,
                      12.0107, 28.0855, 55.834, 100.39, 131.93, 40.096])           # nsc:42
GAMMA_SPECIE = np.array([7. / 5, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 15.6354113,
                         4.913, 1.0125, 2.364, 3.02, 10., 10., 10.])               # nsc:47
MINERAL_DENSITIES = np.array([1.e19] * 7 + [3320, 2260, 2266, 2329, 7870, 3250, 3250., 3166.])  # nsc:44
MRN_CONSTANTS = np.array([50e-10, 5000e-10])                                        # nsc:52
W6_C = 315. / (64. * np.pi)        # poly6 normalisation, nsc:588


# ==========================================================================================
# neighbour search                                                           nsc:541-552
# ==========================================================================================
def neighbors(points, dist, n_neigh, eps=0.1, workers=1):
    """(idx (N,K) int64, None, dist (N,K), nontriv (N,), h (N,)).  nsc:541-552.

    Same third-party call as the reference (scipy cKDTree.query, p=2,
    distance_upper_bound=dist).  eps=0.1 is the reference's setting; eps=0 gives the
    exact kNN that the HIP search implements.  Missing neighbours: idx = N, dist inf -> 0.
    """
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(points, dtype=np.float64)
    tree = cKDTree(pts)
    d, i = tree.query(pts, n_neigh, eps, 2, dist, workers=workers)
    if n_neigh == 1:
        d, i = d[:, None], i[:, None]
    nontriv = np.sum(d < np.inf, axis=1)
    d = np.where(d == np.inf, 0., d)
    h = np.max(d, axis=1)
    return i.astype(np.int64), None, d, nontriv, h


def neighbors_bruteforce(points, dist, n_neigh):
    """O(N^2) exact kNN for small N: independent check of the tree search.
    Ties broken by (distance, index).  Distances computed as sqrt(dx^2+dy^2+dz^2)."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = len(pts)
    dx = pts[:, None, :] - pts[None, :, :]
    d2 = dx[..., 0] ** 2 + dx[..., 1] ** 2 + dx[..., 2] ** 2
    order = np.lexsort((np.broadcast_to(np.arange(n), (n, n)), d2), axis=1)[:, :n_neigh]
    d = np.sqrt(np.take_along_axis(d2, order, axis=1))
    miss = d > dist
    idx = np.where(miss, n, order).astype(np.int64)
    if idx.shape[1] < n_neigh:
        pad = n_neigh - idx.shape[1]
        idx = np.concatenate([idx, np.full((n, pad), n, np.int64)], axis=1)
        d = np.concatenate([d, np.zeros((n, pad))], axis=1)
        miss = np.concatenate([miss, np.ones((n, pad), bool)], axis=1)
    d = np.where(miss, 0., d)
    return idx, None, d, np.sum(~miss, axis=1), np.max(d, axis=1)


# ==========================================================================================
# vectorised SPH sums                                                        nsc:556-671
# ==========================================================================================
def _pair_geometry(nb, pts, vel, h, lo, hi, clip_grad=False):
    """Pair geometry for rows lo:hi.  Deltas are relative to nb[i,0] (nsc:580-581)."""
    n = len(pts)
    nbc = nb[lo:hi]
    valid = nbc < n                                   # idx == N: missing (contributes zero)
    j = np.where(valid, nbc, 0)
    i0 = j[:, 0]
    dx = pts[j] - pts[i0][:, None, :]                 # (n,K,3)
    dv = vel[j] - vel[i0][:, None, :]
    r2 = dx[..., 0] ** 2 + dx[..., 1] ** 2 + dx[..., 2] ** 2
    r = np.sqrt(r2)                                   # nsc:586
    hj = np.where(valid, h[j], 0.)                    # nsc:587
    hi_ = h[lo:hi][:, None]
    with np.errstate(all="ignore"):
        qj = hj * hj - r * r
        W = W6_C / hj ** 9 * qj ** 3                  # nsc:588
        W = np.where(W < 0, 0., W)                    # nsc:589
        cb = -6. * W6_C / hj ** 9 * qj ** 2           # nsc:591 (not clipped)
        if clip_grad:                                 # physics option: as the loop forms, nsc:689
            cb = np.where(qj > 0, cb, 0.)
        qi = hi_ * hi_ - r * r
        ca = -6. * W6_C / hi_ ** 9 * qi ** 2          # nsc:592
    W = np.where(valid, W, 0.)
    cb = np.where(valid, cb, 0.)
    ca = np.where(valid, ca, 0.)
    return j, valid, dx, dv, r2, hj, W, cb, ca


def hydro_update(neighbor, points, mass, sizes, f_un, particle_type, T, mu_array, gamma_array,
                 velocities, chunk=32768, return_intermediates=False, rho_in=None, Bj_in=None,
                 clip_grad=False):
    """Restatement of nsc:556-671 with the axis repair of SURVEY F5 (Pi_i = sum_k pi_ik).

    Returns (hydro_accel (N,3), visc_accel (N,3), visc_heat (N,), density (N,),
    num_density (N,), f_un_neighbor (S,N), dust_density (N,)) - nsc:671 - with the
    reference's sign convention (+grad P / rho).  Rows with idx == N contribute 0.
    rho_in / Bj_in replace the neighbour-side rho_j (nsc:646) and m Pi_j [t==0] (nsc:651) - used
    by the domain-decomposition tests, where ghost rows carry values computed by their owners.
    """
    nb = np.asarray(neighbor).astype(np.int64)
    pts = np.asarray(points, dtype=np.float64)
    vel = np.asarray(velocities, dtype=np.float64)
    m = np.asarray(mass, dtype=np.float64)
    h = np.asarray(sizes, dtype=np.float64)
    fu = np.asarray(f_un, dtype=np.float64)
    pt = np.asarray(particle_type, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    mu = np.asarray(mu_array, dtype=np.float64)
    gam = np.asarray(gamma_array, dtype=np.float64)
    n, K = nb.shape
    S = fu.shape[1]
    gas = (pt == 0.).astype(np.float64)
    dust = (pt == 2.).astype(np.float64)
    # per-particle factors, evaluated in the reference's order of operations
    A = m / mu / AMU * K_B * T * gas                    # pressure weight       nsc:615
    Nw = m / mu / AMU * gas                             # species weight        nsc:626
    with np.errstate(all="ignore"):
        cs = np.sqrt(gam * K_B * T / mu / AMU * gas)    # nsc:647 (neighbour form)
        cs_i = np.sqrt(gam * K_B * T / (mu * AMU) * gas)  # nsc:647 (own form)

    rho = np.zeros(n); rho_d = np.zeros(n); nden = np.zeros(n)
    G = np.zeros((n, 3)); F = np.zeros((S, n))
    inter = {}
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        j, valid, dx, dv, r2, hj, W, cb, ca = _pair_geometry(nb, pts, vel, h, lo, hi, clip_grad)
        mg = np.where(valid, (m * gas)[j], 0.)
        rho[lo:hi] = np.sum(mg * W, axis=1)                                  # nsc:605
        rho_d[lo:hi] = np.sum(np.where(valid, (m * dust)[j], 0.) * W, axis=1)  # nsc:606
        nden[lo:hi] = np.sum(mg * W / AMU / mu[j], axis=1)                   # nsc:607
        term = (np.where(valid, A[j], 0.) * cb)[..., None] * dx \
            + (A[lo:hi][:, None] * ca)[..., None] * dx
        G[lo:hi] = -np.sum(term / 2., axis=1)                                # nsc:615
        if return_intermediates:
            inter.setdefault("G_abs_terms", []).append(np.sum(np.abs(term / 2.), axis=1))
        wN = np.where(valid, Nw[j], 0.) * W
        for s in range(S):
            F[s, lo:hi] = np.sum(wN * fu[j, s], axis=1)                      # nsc:626
        if return_intermediates:
            inter.setdefault("W6_kernel", []).append(W)
            inter.setdefault("distances", []).append(np.sqrt(r2))
            inter.setdefault("neigh_sizes", []).append(hj)
    with np.errstate(all="ignore"):
        hydro_accel = G / rho[:, None]                                        # nsc:619

    # --- viscosity: Pi_i (axis-0 repair of nsc:649) ------------------------------------------
    rho_out = rho
    if rho_in is not None:
        rho = np.asarray(rho_in, dtype=np.float64)
    Pi = np.zeros(n)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        j, valid, dx, dv, r2, hj, W, cb, ca = _pair_geometry(nb, pts, vel, h, lo, hi, clip_grad)
        with np.errstate(all="ignore"):
            w = np.sum(dv * dx, axis=2) / np.sqrt(r2 + 0.01 * hj ** 2)       # nsc:643
            w = np.where(w > 0., 0., w)                                      # nsc:644
            rho_ab = (np.where(valid, rho[j], 0.) + rho[lo:hi][:, None]) / 2.  # nsc:646
            c_ab = 0.5 * (np.where(valid, cs[j], 0.) + cs_i[lo:hi][:, None])   # nsc:647
            pi = -1. / 2. * (c_ab * 2 - 3 * w) * w / rho_ab                  # nsc:649
        pi = np.where(valid, pi, 0.)
        Pi[lo:hi] = np.sum(pi, axis=1)
        if return_intermediates:
            inter.setdefault("w_ab", []).append(w)
            inter.setdefault("rho_avg_ab", []).append(rho_ab)
            inter.setdefault("c_sound_ab", []).append(c_ab)

    visc_accel = np.zeros((n, 3)); visc_heat = np.zeros(n)
    Bj = m * Pi * gas                                                         # nsc:651
    Bj_own = Bj
    if Bj_in is not None:
        Bj = np.asarray(Bj_in, dtype=np.float64)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        j, valid, dx, dv, r2, hj, W, cb, ca = _pair_geometry(nb, pts, vel, h, lo, hi, clip_grad)
        B = ((np.where(valid, Bj[j], 0.) * cb)[..., None] * dx
             + (Bj_own[lo:hi][:, None] * ca)[..., None] * dx) / 2.
        visc_accel[lo:hi] = -np.sum(B, axis=1)                                # nsc:651-652
        if return_intermediates:
            inter.setdefault("visc_abs_terms", []).append(np.sum(np.abs(B), axis=1))
        visc_heat[lo:hi] = np.sum(B * dv, axis=(1, 2)) * m[lo:hi] / 2.        # nsc:653-654
    out = (hydro_accel, visc_accel, visc_heat, rho_out, nden, F, rho_d)
    if return_intermediates:
        inter = {k_: np.concatenate(v, axis=0) for k_, v in inter.items()}
        inter["Pi"] = Pi
        inter["Bw"] = Bj_own
        inter["pressure_grad_symmetrized"] = G.T
        return out, inter
    return out


# ==========================================================================================
# loop forms                                                                 nsc:673-816
# ==========================================================================================
def _loop_geometry(points, neighbor):
    pts = np.asarray(points, dtype=np.float64)
    nb = np.asarray(neighbor).astype(np.int64)
    dx = pts[nb] - pts[:, None, :]                     # relative to particle i itself
    r2 = np.sum(dx ** 2, axis=2)
    return pts, nb, dx, r2


def _weigh2(r2, m, d):
    """m * W6(r, h(m)), h(m) = (m/m_0)^(1/3) d.   nsc:673-676 (not clipped)."""
    return m * 315 * (M_0 / m) ** 3 * ((m / M_0) ** (2. / 3.) * d ** 2 - r2) ** 3 / (64 * np.pi * d ** 9)


def _weigh2_dust(r2, m, ds):
    """nsc:678-681."""
    return m * 315 * (ds ** 2 - r2) ** 3 / (64 * np.pi * ds ** 9)


def _grad_weight(dx, r2, m, d, ptype_nb):
    """nsc:683-690: clipped gradient, gas neighbours only, no mass factor."""
    q = (m / M_0) ** (2. / 3.) * d ** 2 - r2
    c = -315 * 6 * (M_0 / m) ** 3 / (64 * np.pi * d ** 9) * q ** 2 * (ptype_nb == 0)
    return np.nan_to_num((c * (q > 0))[..., None] * dx)


def density(points, mass, particle_type, neighbor, d):
    """nsc:693-702."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    m = np.asarray(mass, dtype=np.float64)[nb]
    rho = _weigh2(r2, m, d) * (np.asarray(particle_type)[nb] == 0)
    return np.sum(np.where(rho > 0, rho, 0.), axis=1)


def dust_density(points, mass, neighbor, particle_type, sizes, d=None):
    """nsc:704-717."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    m = np.asarray(mass, dtype=np.float64)[nb]
    ds = np.asarray(sizes, dtype=np.float64)[nb]
    with np.errstate(all="ignore"):
        rho = _weigh2_dust(r2, m, ds) * (np.asarray(particle_type)[nb] == 2)
    return np.sum(np.where(rho > 0, rho, 0.), axis=1)


def num_dens(mass, points, mu_array, neighbor, d):
    """nsc:744-753 (no type mask; m_h, not amu)."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    m = np.asarray(mass, dtype=np.float64)[nb]
    nd = _weigh2(r2, m, d) / (np.asarray(mu_array, dtype=np.float64)[nb] * M_H)
    return np.sum(np.where(nd > 0, nd, 0.), axis=1)


def del_pressure(points, mass, particle_type, neighbor, E_internal, gamma_array, d):
    """nsc:755-774."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    pt = np.asarray(particle_type)
    m = np.asarray(mass, dtype=np.float64)[nb]
    E = np.asarray(E_internal, dtype=np.float64)
    gw = _grad_weight(dx, r2, m, d, pt[nb])
    fac = 0.5 * (E[nb] + E[:, None]) / np.asarray(gamma_array, dtype=np.float64)[nb]
    out = np.sum(gw * fac[..., None], axis=1)
    return out * (pt == 0)[:, None]


def artificial_viscosity(neighbor, points, particle_type, sizes, mass, densities, velocities, T,
                         gamma_array, mu_array, d):
    """nsc:788-816."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    pt = np.asarray(particle_type)
    m = np.asarray(mass, dtype=np.float64)
    vel = np.asarray(velocities, dtype=np.float64)
    rho = np.asarray(densities, dtype=np.float64)
    with np.errstate(all="ignore"):
        css = np.nan_to_num((np.asarray(gamma_array) * K_B * np.asarray(T)
                             / (np.asarray(mu_array) * AMU)) ** 0.5)
        dv = vel[nb] - vel[:, None, :]
        w = np.sum(dv * dx, axis=2) / r2 ** 0.5
        w = np.where(w > 0, 0., w)
        w = np.nan_to_num(w)
        vsig = css[nb] + css[:, None] - 3 * w
        rho_ij = (rho[nb] + rho[:, None]) / 2.
        PI = -1. / 2. * vsig * w / rho_ij
        gw = _grad_weight(dx, r2, m[:, None], d, pt[nb])
        mbar = (m[nb] + m[:, None]) / 2.
        accel_ij = (mbar * PI)[..., None] * gw
        heat_ij = 0.5 * mbar * PI * np.sum(dv * gw, axis=2)
    gasnb = pt[nb] == 0
    accel = np.sum(np.where(gasnb[..., None], accel_ij, 0.), axis=1)
    heat = np.sum(np.where(gasnb, heat_ij, 0.), axis=1)
    act = (pt == 0) & (np.sum(gasnb, axis=1) > 0)
    return accel * act[:, None], heat * act


def crossing_time(neighbor, velocities, sizes, particle_type, dt_0=DT_0):
    """nsc:776-786."""
    nb = np.asarray(neighbor).astype(np.int64)
    vel = np.asarray(velocities, dtype=np.float64)
    rel = np.max(np.sum((vel[nb] - vel[:, None, :]) ** 2, axis=2), axis=1) ** 0.5
    with np.errstate(all="ignore"):
        ct = np.nan_to_num(np.asarray(sizes, dtype=np.float64) / rel) * (np.asarray(particle_type) == 0)
    nz = ct[ct != 0]
    if len(nz) == 0:
        return dt_0 / 10.
    return float(np.min(nz)) + 0.0001


def grain_mass():
    """nsc:76-79."""
    a = MRN_CONSTANTS
    return MINERAL_DENSITIES * -(a[1] ** 0.5 - a[0] ** 0.5) / (a[1] ** -2.5 - a[0] ** -2.5) \
        * (4. / 5.) * 4 * np.pi / 3.


def sigma_effective():
    """nsc:69-74."""
    a = MRN_CONSTANTS
    return MU_SPECIE * AMU / MINERAL_DENSITIES * (3. / 4.) \
        * -(a[1] ** -0.5 - a[0] ** -0.5) / (a[1] ** 0.5 - a[0] ** 0.5)


def net_impulse(points, mass, sizes, velocities, particle_type, neighbor, f_un, d=None):
    """nsc:719-742: dust -> gas drag with scatter-added reaction."""
    pts, nb, dx, r2 = _loop_geometry(points, neighbor)
    n = len(pts)
    pt = np.asarray(particle_type)
    fu = np.asarray(f_un, dtype=np.float64)
    vel = np.asarray(velocities, dtype=np.float64)
    meff, seff = grain_mass(), sigma_effective()
    mgm = fu @ meff
    mcs = fu @ seff
    dv = vel[nb] - vel[:, None, :]
    with np.errstate(all="ignore"):
        wf = _weigh2_dust(r2, np.asarray(mass, dtype=np.float64)[nb], np.asarray(sizes, dtype=np.float64)[nb])
        coef = wf / mgm[nb] * mcs[nb] * np.sum(dv ** 2, axis=2) ** 0.5 * (pt[nb] == 2) * (wf > 0)
        acc = coef[..., None] * dv
    onto = np.sum(acc, axis=1)
    react = np.zeros((n, 3))
    notself = (nb != np.arange(n)[:, None])
    np.add.at(react, nb.ravel(), (-(acc * notself[..., None])).reshape(-1, 3))
    return onto, react


# ==========================================================================================
# integrator + time-step control (drv:222-238, drv:460-491)
# Pinned: tests/golden/make_golden_driver.py executes exactly those two statement blocks of the
# reference driver (taken by line range; the script as a whole is not runnable) on seeded
# arrays; tests/test_oracle_golden.py checks `timestep`, `clamp_state`, `assemble_loop` and `leapfrog`
# against what the blocks left behind, bit for bit.
# ==========================================================================================
MAX_AGE = 3e7 * YEAR               # drv:78


def timestep(ct, first, dt_0=DT_0, max_age=MAX_AGE):
    """drv:223-229."""
    dt = dt_0 / 10 if first else max(dt_0 / 5., min(dt_0 * 2., ct))
    if ct > max_age:
        dt = max_age / 100.
    return dt


def clamp_state(points, velocities):
    """drv:233-238.  (The two velocity assignments at drv:235-236 test the positions AFTER they were
    clamped, so they never select anything: positions beyond the limit are set onto it, NaN -> 0.)"""
    lim = 1e11 * AU
    p = np.nan_to_num(np.clip(points, -lim, lim))
    return p, np.nan_to_num(velocities)


def assemble_loop(delp, rho, rho_d, particle_type, av_accel, drag=None):
    """drv:460-463,473: pressure acceleration and the viscous acceleration before the limiter, from the
    loop forms' outputs.  drag = (onto_gas (N,3), reaction (N,3)) or None."""
    gas = (np.asarray(particle_type) == 0).astype('float')
    with np.errstate(all="ignore"):
        pressure_accel = np.nan_to_num((delp.T / rho * gas).T)                       # drv:460
        if drag is None:
            return pressure_accel, av_accel
        drag_gas = np.nan_to_num((drag[0].T * rho_d / rho * gas).T)                  # drv:462
        drag_dust = np.nan_to_num(drag[1])                                           # drv:463
        return pressure_accel, drag_gas + drag_dust + av_accel                       # drv:473


def leapfrog(p, v, old, E_internal, mass, mu_array, gamma_array, pressure_accel, visc, heat, dt, grav_accel=None):
    """drv:472-491: viscous limiter, total acceleration, leapfrog, energy, temperature.
    Returns (points, velocities, total_accel, E_internal, T)."""
    with np.errstate(all="ignore"):
        lim = (np.sum(v ** 2, axis=1) ** 0.5 - np.sum(visc ** 2, axis=1) ** 0.5 * dt) < 0   # drv:475
        visc = np.where(lim[:, None], -v / dt, visc)
        if grav_accel is not None:
            total = grav_accel + pressure_accel + visc                               # drv:477
        else:
            total = pressure_accel + visc
        pn = p + ((total * dt ** 2) / 2. + v * dt)                                   # drv:481
        if np.shape(total) == np.shape(old):
            vn = v + (total + old) / 2. * dt                                         # drv:482-483
        else:
            vn = v + total * dt                                                      # drv:484-485
        E = np.nan_to_num(E_internal) + np.nan_to_num(heat * dt)                     # drv:490
        T = np.nan_to_num(E * (mu_array * M_H) / (gamma_array * mass * K_B))         # drv:491
    return pn, vn, total, E, T


def integrate(p, v, old, E_internal, mass, mu_array, gamma_array, particle_type, ha, va, vh, dt,
              grav_accel=None, drag=None):
    """The update on `hydro_update`'s sums: acceleration in the physical sign (SURVEY Q2:
    a = -(hydro_accel + visc_accel), gas only), then drv:472-491.
    Returns (points, velocities, total_accel, E_internal, T)."""
    gas = (np.asarray(particle_type) == 0.)[:, None]
    with np.errstate(all="ignore"):
        pressure_accel = np.nan_to_num(-ha * gas)                                    # cf. drv:460
        visc = np.nan_to_num(-va * gas)
        if drag is not None:                                                         # drv:462-463,473
            onto, react, rho, rho_d = drag
            visc = np.nan_to_num(onto * (rho_d / rho)[:, None] * gas) + np.nan_to_num(react) + visc
    return leapfrog(p, v, old, E_internal, mass, mu_array, gamma_array, pressure_accel, visc, vh, dt, grav_accel)


def step_loop(state, d, n_neigh=40, dist=np.inf, eps=0.0, first=False, fixed_dt=0.0, with_drag=False,
              grav_accel=None, workers=1):
    """One pass of the reference's time loop as it is written (drv:222-238, 437, 451-491): the LOOP
    forms on the step's neighbour list, global d."""
    s = dict(state)
    p, v = clamp_state(s["points"], s["velocities"])
    m, pt = s["mass"], s["particle_type"]
    nb, _, _, _, h = neighbors(p, dist, n_neigh, eps, workers=workers)               # drv:437
    ct = crossing_time(nb, v, h, pt)                                                  # drv:222
    dt = fixed_dt if fixed_dt > 0 else timestep(ct, first)
    rho = density(p, m, pt, nb, d)                                                    # drv:451
    rho_d = dust_density(p, m, nb, pt, h)                                             # drv:452
    nden = num_dens(m, p, s["mu_array"], nb, d)                                       # drv:453
    delp = del_pressure(p, m, pt, nb, s["E_internal"], s["gamma_array"], d)           # drv:456
    av = artificial_viscosity(nb, p, pt, h, m, rho, v, s["T"], s["gamma_array"], s["mu_array"], d)   # drv:458
    drag = net_impulse(p, m, h, v, pt, nb, s["f_un"]) if with_drag else None          # drv:455
    pressure_accel, visc = assemble_loop(delp, rho, rho_d, pt, av[0], drag)
    pn, vnew, total, E, T = leapfrog(p, v, s["total_accel"], s["E_internal"], m, s["mu_array"], s["gamma_array"],
                                     pressure_accel, visc, av[1], dt, grav_accel)
    s.update(points=pn, velocities=vnew, total_accel=total, E_internal=E, T=T, dt=dt, sizes=h, densities=rho,
             num_densities=nden, dust_densities=rho_d, neighbor=nb)
    return s


G_NEWTON = 6.67430e-11        # scipy.constants.G (nsc:21)


def gravity_direct(points, mass, softening, G=G_NEWTON, chunk=512):
    """The sum the reference's tree gravity (nsc:252-415) approximates: every particle as its own
    monopole, Plummer-softened as at nsc:385:  G m_j (x_j - x_i) / (|x_j - x_i|^2 + eps^2)^(3/2)."""
    p = np.asarray(points, dtype=np.float64)
    m = np.asarray(mass, dtype=np.float64)
    out = np.zeros_like(p)
    e2 = float(softening) ** 2
    for a in range(0, len(p), chunk):
        d = p[None, :, :] - p[a:a + chunk, None, :]                    # (c, n, 3)
        r2 = np.sum(d * d, axis=2) + e2
        with np.errstate(all="ignore"):
            w = np.where(r2 > 0, m[None, :] / (r2 * np.sqrt(r2)), 0.0)
        out[a:a + chunk] = G * np.sum(w[:, :, None] * d, axis=1)
    return out


def step(state, n_neigh=40, dist=np.inf, eps=0.0, first=False, grav_accel=None, workers=1,
         fixed_dt=0.0, with_drag=False, with_gravity=False, grav_G=None, clip_grad=False):
    """One pass of the hot path: search -> dt -> sums -> leapfrog update.

    state: dict with points, velocities, mass, particle_type, f_un, T, mu_array, gamma_array,
    E_internal, total_accel.  Returns the new state (+ 'dt', 'sizes', 'densities').
    Physical sign: a = -(hydro_accel + visc_accel) [+ grav]   (SURVEY Appendix B, Q2).
    """
    s = dict(state)
    p, v = clamp_state(s["points"], s["velocities"])
    nb, _, _, _, h = neighbors(p, dist, n_neigh, eps, workers=workers)
    ct = crossing_time(nb, v, h, s["particle_type"])
    dt = fixed_dt if fixed_dt > 0 else timestep(ct, first)
    fu = s.get("f_un")
    if fu is None:                       # light ICs do not carry the (N,15) composition
        fu = np.ones((len(p), 1))
    ha, va, vh, rho, nden, F, rho_d = hydro_update(nb, p, s["mass"], h, fu, s["particle_type"],
                                                   s["T"], s["mu_array"], s["gamma_array"], v, clip_grad=clip_grad)
    if with_gravity:                                                           # drv:448-449, nsc:358
        grav_accel = gravity_direct(p, s["mass"], np.median(h), G=G_NEWTON if grav_G is None else grav_G)
    drag = None
    if with_drag:                                                              # drv:455
        onto, react = net_impulse(p, s["mass"], h, v, s["particle_type"], nb, s["f_un"])
        drag = (onto, react, rho, rho_d)
    p, v, total, E, T = integrate(p, v, s["total_accel"], s["E_internal"], s["mass"], s["mu_array"],
                                  s["gamma_array"], s["particle_type"], ha, va, vh, dt, grav_accel, drag)
    s.update(points=p, velocities=v, total_accel=total, E_internal=E, T=T, dt=dt, sizes=h,
             densities=rho, num_densities=nden, neighbor=nb)
    return s
