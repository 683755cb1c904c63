"""Seeded initial conditions of the BASELINE configs (np.random.RandomState,
seed = 12345 + config number).  Array conventions follow sph/code_running.py:114-177.

The named ICs do not exist in the reference; they are synthesised here:
  C1 uniform sphere        R = 0.625e6 AU, m = 0.4 Msun/715, gas, T ~ U[10,20) K, v ~ N(0, 1 km/s)
  C2 polytrope-like sphere n(r) ~ 1/((r/2e7 AU)^4 + 1), r <= 5e7 AU: the initial radial profile of
                           imf/lane_emden_modified.py:317-343 sampled by inverse CDF
  C3 Sedov-Taylor blast    uniform cold sphere + 1e44 J at the centre, 73.6 % kinetic (radial kick,
                           nsc:1193-1201) / 26.4 % thermal (nsc:909), dealt to the innermost particles
"""
import numpy as np

AU = 149597870700.0
SOLAR = 1.989e30
K_B = 1.380649e-23
AMU = 1.66053906892e-27
M_H = 1.0008 * AMU
MU_SPECIE = np.array([2.0158, 4.0026, 1.0079, 1.0074, 4.0021, 4.0016, 0.0005, 140.69, 60.08, 12.0107,
                      28.0855, 55.834, 100.39, 131.93, 40.096])
GAMMA_SPECIE = np.array([7. / 5, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 5. / 3, 15.6354113, 4.913,
                         1.0125, 2.364, 3.02, 10., 10., 10.])
F_GAS = np.array([.86, .14] + [0.] * 13)
F_DUST = np.array([0.] * 7 + [0.125] * 8)        # drv:84,114 (dust_base, normalised)
DUST_MASS = 0.05000000001 / 715. * SOLAR         # drv:76


def _finish(pts, vel, mass, T, ptype=None, f_un=None, light=False):
    """light=True: do not materialise the (N,15) composition (all-gas ICs of many millions of
    particles); mu/gamma are the same expressions evaluated on the single composition row."""
    n = len(pts)
    ptype = np.zeros(n) if ptype is None else ptype
    if f_un is None and light:
        row = F_GAS[None, :]
        mu = np.full(n, (np.sum(row * MU_SPECIE, axis=1) / np.sum(row, axis=1))[0])
        gam = np.full(n, (np.sum(row * GAMMA_SPECIE, axis=1) / np.sum(row, axis=1))[0])
        E = gam * mass * K_B * T / (mu * M_H)
        return dict(points=np.ascontiguousarray(pts), velocities=np.ascontiguousarray(vel), mass=mass,
                    particle_type=ptype, f_un=None, T=T, mu_array=mu, gamma_array=gam, E_internal=E,
                    total_accel=np.zeros((n, 3)))
    if f_un is None:
        f_un = np.tile(F_GAS, (n, 1))
    mu = np.sum(f_un * MU_SPECIE, axis=1) / np.sum(f_un, axis=1)        # drv:162
    gam = np.sum(f_un * GAMMA_SPECIE, axis=1) / np.sum(f_un, axis=1)    # drv:163
    E = gam * mass * K_B * T / (mu * M_H)                               # drv:166
    return dict(points=np.ascontiguousarray(pts), velocities=np.ascontiguousarray(vel), mass=mass,
                particle_type=ptype, f_un=f_un, T=T, mu_array=mu, gamma_array=gam, E_internal=E,
                total_accel=np.zeros((n, 3)))


def _unit_vectors(rs, n):
    u = rs.normal(size=(n, 3))
    return u / np.linalg.norm(u, axis=1)[:, None]


def uniform_sphere(n, seed=12346, radius=0.625e6 * AU, sigma_v=1000., light=False, size_scale=1.0):
    """C1.  size_scale multiplies every length of the IC (density ~ n / size_scale^3)."""
    rs = np.random.RandomState(seed)
    pts = _unit_vectors(rs, n) * (rs.rand(n) ** (1. / 3.))[:, None] * (radius * size_scale)
    mass = np.full(n, 0.4 * SOLAR / 715.)
    T = 10. * (1. + rs.rand(n))
    vel = rs.normal(size=(n, 3)) * sigma_v
    return _finish(pts, vel, mass, T, light=light)


def polytrope_sphere(n, seed=12347, a=2e7 * AU, rmax=5e7 * AU, sigma_v=1000., light=False, size_scale=1.0):
    """C2: radii by inverse CDF of M(<r) ~ int r^2/((r/a)^4+1) dr on the reference's radial grid
    r = sqrt(linspace(0, 25e14, 2001)) AU (imf/lane_emden_modified.py:317-343)."""
    rs = np.random.RandomState(seed)
    a, rmax = a * size_scale, rmax * size_scale
    r_grid = np.sqrt(np.linspace(0., (rmax / AU) ** 2, 2001)) * AU
    dens = 1. / ((r_grid / a) ** 4 + 1.)
    shell = 0.5 * (dens[1:] * r_grid[1:] ** 2 + dens[:-1] * r_grid[:-1] ** 2) * np.diff(r_grid)
    cdf = np.concatenate([[0.], np.cumsum(shell)])
    cdf /= cdf[-1]
    r = np.interp(rs.rand(n), cdf, r_grid)
    pts = _unit_vectors(rs, n) * r[:, None]
    # equal-mass particles; total mass = mean density 1e-18 kg/m^3-scale cloud of the 1-D model
    mass = np.full(n, 0.4 * SOLAR / 715.)
    T = 10. * (1. + rs.rand(n))
    vel = rs.normal(size=(n, 3)) * sigma_v
    return _finish(pts, vel, mass, T, light=light)


def sedov_sphere(n, seed=12348, radius=0.625e6 * AU, energy=1e44, kick_mass=194.28 * SOLAR, light=False,
                 size_scale=1.0):
    """C3: cold uniform gas; the innermost `kick_mass` of gas receives 73.6 % of `energy` as a
    radial kick and 26.4 % as heat."""
    rs = np.random.RandomState(seed)
    pts = _unit_vectors(rs, n) * (rs.rand(n) ** (1. / 3.))[:, None] * (radius * size_scale)
    mass = np.full(n, 0.4 * SOLAR / 715.)
    T = 10. * (1. + rs.rand(n))
    vel = rs.normal(size=(n, 3)) * 100.
    r = np.linalg.norm(pts, axis=1)
    order = np.argsort(r)
    nk = max(int(round(kick_mass / mass[0])), 32)
    nk = min(nk, n // 8)
    inner = order[:nk]
    ek = 0.736 * energy
    vk = np.sqrt(2. * ek / np.sum(mass[inner]))
    rhat = pts[inner] / np.maximum(r[inner], 1e-300)[:, None]
    vel[inner] += vk * rhat
    st = _finish(pts, vel, mass, T, light=light)
    eth = 0.264 * energy / nk
    st["E_internal"][inner] += eth
    st["T"][inner] = st["E_internal"][inner] * (st["mu_array"][inner] * M_H) / \
        (st["gamma_array"][inner] * mass[inner] * K_B)                   # drv:491
    return st


def uniform_cube(n, seed=12349, side=1.25e6 * AU, sigma_v=1000., light=False, size_scale=1.0, box=None):
    """C4: the reference's own IC (sph/code_running.py:62,132).  box = (lo, hi): the n particles fill that part of the
    cube instead (one rank's brick of a decomposed run, seeded per brick: ics.cube_slab)."""
    rs = np.random.RandomState(seed)
    if box is not None:
        blo, bhi = np.asarray(box[0], dtype=np.float64), np.asarray(box[1], dtype=np.float64)
        pts = blo[None, :] + rs.rand(n, 3) * (bhi - blo)[None, :]
    else:
        pts = (rs.rand(n, 3) - 0.5) * (side * size_scale)
    mass = np.full(n, 0.4 * SOLAR / 715.)
    T = 10. * (1. + rs.rand(n))
    vel = rs.normal(size=(n, 3)) * sigma_v
    return _finish(pts, vel, mass, T, light=light)


def bench_size_scale(n_total):
    """Length scale of the benchmark cloud.  1 at the BASELINE size (1e6 particles: the reference's IC as
    it is).  Beyond it the cloud is enlarged by (N/1e6)^(1/3) x 16: the reference's scheme is unstable
    (unclipped neighbour gradient, nsc:591; first dt up to 2 dt_0, drv:226) and the more particles a cloud
    has the worse its worst local configuration - at constant density a 2e6..8e6-particle polytrope
    diverges within 3 steps (max|v| 1e22 m/s), diluted x4 the 8e6 one still does after 12; at x16 the
    fastest particle drifts ~0.3 h per step, as in the 1e6 case, and 4e6 / 8e6 particles stay sane for
    30+ steps (measured on one MI355X).  The work per particle-step does not depend on the scale."""
    if n_total <= 1_000_000:
        return 1.0
    return 16.0 * (n_total / 1e6) ** (1. / 3.)


def cfl_dt(state, k=40, courant=0.25):
    """A Courant-limited fixed step for violent ICs (the reference's dt >= dt_0/5 floor of
    sph/code_running.py:226 is far above the crossing time of a 1e44 J blast at this scale):
    courant * (mean kNN radius of a uniform fill) / (max speed + max sound speed)."""
    pts = state["points"]
    n = len(pts)
    r = np.linalg.norm(pts - pts.mean(axis=0), axis=1).max()
    hbar = (k / n) ** (1. / 3.) * r
    cs = np.sqrt(state["gamma_array"] * K_B * state["T"] / (state["mu_array"] * AMU)).max()
    vmax = np.linalg.norm(state["velocities"], axis=1).max()
    return float(courant * hbar / (vmax + cs))


def loop_d(state, k=40, sample=200000):
    """The driver's global d (drv:67-68) for a synthetic IC: chosen so that the loop forms' mass-derived smoothing
    length h(m) = (m/m_0)^(1/3) d (nsc:675) is the median kNN radius of the cloud (estimated on a subsample)."""
    from scipy.spatial import cKDTree
    pts = state["points"]
    n = len(pts)
    sub = pts[:: max(1, n // sample)]
    dd = cKDTree(sub).query(sub, k=min(k, len(sub)))[0]                    # (k = 1: a 1-d array of zeros - the particle itself)
    hs = np.asarray(dd).reshape(len(sub), -1)[:, -1] * (len(sub) / n) ** (1. / 3.)
    return float(np.median(hs) / np.median((state["mass"] / 10 ** 1.5 / SOLAR) ** (1. / 3.)))


def dusty_sphere(n, seed=12350, dust_frac=0.05, **kw):
    """C1 with a fraction of the particles turned into dust SPH particles (particle_type 2,
    drv:127,150-152): exercises dust density and the gas-dust drag (nsc:719-742)."""
    s = uniform_sphere(n, seed=seed, **kw)
    rs = np.random.RandomState(seed + 1)
    dust = rs.permutation(n)[:int(round(dust_frac * n))]
    ptype = np.zeros(n)
    ptype[dust] = 2.
    mass = s["mass"].copy()
    mass[dust] = DUST_MASS
    f_un = np.where((ptype == 2.)[:, None], F_DUST[None, :], F_GAS[None, :])
    return _finish(s["points"], s["velocities"], mass, s["T"], ptype=ptype, f_un=f_un)


def cube_slab(workload, n_rank, rank, box_lo, box_hi, light=False):
    """One rank's share of a decomposed uniform_cube / two_phase run WITHOUT building the global state on every rank: n_rank
    particles uniform in the rank's own brick (multigpu.brick_regions), every other field drawn as the global IC draws it,
    the stream seeded by (workload seed, rank).  The union over the ranks is a uniform fill of the cube with exactly
    world x n_rank particles - the same ensemble as WORKLOADS[workload](world * n_rank), not the same sample."""
    if workload == "uniform_cube":
        return uniform_cube(n_rank, seed=12349 + 1000 * (rank + 1), light=light, box=(box_lo, box_hi))
    if workload == "two_phase":
        return two_phase(n_rank, seed=12351 + 1000 * (rank + 1), box=(box_lo, box_hi))
    raise ValueError("cube_slab: %s does not fill a box" % workload)


def two_phase(n, seed=12351, dust_frac=0.10, **kw):
    """C5: the reference's own two-phase IC (drv:120-152) - gas particles and dust SPH particles (particle_type 2,
    mass DUST_MASS, composition dust_base) thrown into the same uniform cube - with a tenth of the particles dust."""
    kw.pop("light", None)
    s = uniform_cube(n, seed=seed, **kw)
    rs = np.random.RandomState(seed + 1)
    dust = rs.permutation(n)[:int(round(dust_frac * n))]
    ptype = np.zeros(n)
    ptype[dust] = 2.
    mass = s["mass"].copy()
    mass[dust] = DUST_MASS
    f_un = np.where((ptype == 2.)[:, None], F_DUST[None, :], F_GAS[None, :])
    return _finish(s["points"], s["velocities"], mass, s["T"], ptype=ptype, f_un=f_un)


WORKLOADS = {"two_phase": two_phase, "dusty_sphere": dusty_sphere, "uniform_sphere": uniform_sphere, "polytrope": polytrope_sphere, "sedov": sedov_sphere,
             "uniform_cube": uniform_cube}
