#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

CONTAINER-ONLY TOOL.  It reads /root/reference/sph/navier_stokes_cleaned.py
("nsc") at run time, applies a mechanical in-memory Python-2 -> Python-3
transform (tab expansion + lib2to3 `fix_print`), executes the result in a fresh
module object and calls the reference's own functions on seeded inputs.  No
reference source text is stored in this repository or in the fixtures: the
.npz files hold only inputs, expected outputs and captured intermediates.

It is a no-op (exit 0 with a message) when /root/reference is absent, e.g. on
the GPU box.

What is captured:
  * nsc.neighbors(points, dist, K)             nsc:541-552   (eps = 0.1)
  * the same cKDTree queried with eps = 0      (exact-kNN fixture; SciPy, not
    reference code - pins the exact search the HIP path implements)
  * nsc.hydro_update(...) as committed: runs nsc:556-649 and raises IndexError
    at nsc:651; the locals of the failing frame are captured from
    the traceback (unmodified reference arithmetic)
  * nsc.hydro_update(...) with the one-token repair `axis=1 -> axis=0` on the
    artificial_viscosity_Pi line (nsc:649), all 7 outputs
  * loop forms nsc:673-816 with the global `d` injected: density, dust_density,
    num_dens, del_pressure, artificial_viscosity, crossing_time, net_impulse

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/sph/navier_stokes_cleaned.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference(patch_axis):
    """Return the reference module object (py3-transformed in memory)."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    warnings.simplefilter("ignore")
    from lib2to3 import refactor

    src = open(REF).read().expandtabs(8)
    if not src.endswith("\n"):
        src += "\n"
    if patch_axis:
        lines = src.split("\n")
        hits = [i for i, l in enumerate(lines)
                if l.strip().startswith("artificial_viscosity_Pi = np.sum(")]
        assert len(hits) == 1, hits
        assert lines[hits[0]].rstrip().endswith("axis=1)")
        lines[hits[0]] = lines[hits[0]].rstrip()[:-len("axis=1)")] + "axis=0)"
        src = "\n".join(lines)
    tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    src3 = str(tool.refactor_string(src, "nsc"))
    mod = types.ModuleType("nsc_ref_patched" if patch_axis else "nsc_ref")
    exec(compile(src3, mod.__name__, "exec"), mod.__dict__)
    return mod


# ----------------------------------------------------------------------------
# seeded inputs (SI units, shaped like sph/code_running.py:114-177)
# ----------------------------------------------------------------------------
AU = 149597870700.0
SOLAR = 1.989e30
F_GAS = np.array([.86, .14] + [0.] * 13)
F_DUST = np.array([0.] * 7 + [0.125] * 8)


def make_case(name, n, k, seed, shape, dust_frac, star_count, vel_mode):
    rs = np.random.RandomState(seed)
    L = 1.25e6 * AU
    if shape == "cube":
        pts = (rs.rand(n, 3) - 0.5) * L
    elif shape == "sphere":
        u = rs.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1)[:, None]
        pts = u * (rs.rand(n) ** (1. / 3.))[:, None] * (L / 2)
    elif shape == "condensed":
        u = rs.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1)[:, None]
        pts = u * (rs.rand(n) ** 1.2)[:, None] * (L / 2)   # rho ~ r^-0.5 cusp
    else:
        raise ValueError(shape)
    mass = np.full(n, 0.4 * SOLAR / 715.) * (1.0 + 0.25 * rs.rand(n))
    ptype = np.zeros(n)
    nd = int(round(dust_frac * n))
    perm = rs.permutation(n)
    ptype[perm[:nd]] = 2.
    ptype[perm[nd:nd + star_count]] = 1.
    mass[ptype == 2.] = 0.05000000001 / 715. * SOLAR
    T = 10. * (1. + rs.rand(n))
    if vel_mode == "thermal":
        vel = rs.normal(size=(n, 3)) * 1000.
    elif vel_mode == "converging":
        vel = -pts / (L / 2) * 5000. + rs.normal(size=(n, 3)) * 300.
        T = T * (1. + 50. * np.exp(-np.sum(pts ** 2, axis=1) / (0.1 * L) ** 2))
    else:
        raise ValueError(vel_mode)
    f_un = np.where((ptype == 2.)[:, None], F_DUST[None, :], F_GAS[None, :])
    # small per-particle composition scatter so that species sums are non-trivial
    f_un = f_un * (1. + 0.1 * rs.rand(n, 15))
    return dict(name=name, n=n, k=k, points=pts, mass=mass, particle_type=ptype,
                T=T, velocities=vel, f_un=f_un)


CASES = [
    #    name                    n     k  seed  shape       dust  stars vel
    ("cube_gas_n2048_k40",     2048, 40, 101, "cube",      0.00, 0, "thermal"),
    ("sphere_dust_n2048_k40",  2048, 40, 102, "sphere",    0.05, 3, "thermal"),
    ("condensed_n1024_k40",    1024, 40, 103, "condensed", 0.00, 0, "thermal"),
    ("converging_n1024_k16",   1024, 16, 104, "sphere",    0.02, 1, "converging"),
    ("small_n256_k16",          256, 16, 105, "cube",      0.05, 2, "thermal"),
]
FULL_INTERMEDIATES = {"small_n256_k16"}


def run_case(ref, refp, spec):
    c = make_case(*spec)
    n, k = c["n"], c["k"]
    pts, mass, ptype, T, vel, f_un = (c[x] for x in
                                      ("points", "mass", "particle_type", "T", "velocities", "f_un"))
    mu_specie, gamma_specie = ref.mu_specie, ref.gamma
    mu = np.sum(f_un * mu_specie, axis=1) / np.sum(f_un, axis=1)       # code_running.py:162
    gam = np.sum(f_un * gamma_specie, axis=1) / np.sum(f_un, axis=1)   # code_running.py:163
    E = gam * mass * ref.k * T / (mu * ref.m_h)                        # code_running.py:166
    out = dict(points=pts, mass=mass, particle_type=ptype, T=T, velocities=vel, f_un=f_un,
               mu_array=mu, gamma_array=gam, E_internal=E, K=np.int64(k))

    # --- neighbour search -----------------------------------------------------
    dist_bound = 1e30
    nb, kdt, nd, nontriv, h = ref.neighbors(pts, dist_bound, k)
    out.update(nb_idx=nb.astype(np.int32), nb_dist=nd, nb_nontriv=nontriv.astype(np.int32), nb_h=h,
               dist_bound=np.float64(dist_bound))
    de, ie = kdt.query(pts, k, 0.0, 2, dist_bound)      # exact kNN, same SciPy tree
    out.update(exact_idx=ie.astype(np.int32), exact_dist=de)
    # a bounded search that leaves missing neighbours (idx == N, dist inf -> 0)
    rb = float(np.median(h)) * 0.9
    nb2, _, nd2, nt2, h2 = ref.neighbors(pts, rb, k)
    out.update(bounded_r=np.float64(rb), bounded_idx=nb2.astype(np.int32), bounded_dist=nd2,
               bounded_nontriv=nt2.astype(np.int32), bounded_h=h2)

    # --- hydro_update as committed (raises at nsc:651) -> captured locals ------
    args = (nb, pts, mass, h, f_un, ptype, T, mu, gam, vel)
    captured = None
    try:
        ref.hydro_update(*args)
    except IndexError:
        tb = sys.exc_info()[2]
        while tb.tb_next is not None:
            tb = tb.tb_next
        captured = dict(tb.tb_frame.f_locals)
    assert captured is not None, "reference did not raise (n <= k?)"
    for nm in ("density_calc", "dust_density_calc", "num_density_calc",
               "pressure_grad_symmetrized", "hydro_accel", "f_un_neighbor"):
        out["cap_" + nm] = np.array(captured[nm], dtype=np.float64)
    if c["name"] in FULL_INTERMEDIATES:
        for nm in ("delt_x", "delt_vels", "distances", "neigh_sizes", "W6_kernel",
                   "W6_grad_a", "W6_grad_b", "w_ab", "rho_avg_ab", "c_sound_ab"):
            out["cap_" + nm] = np.array(captured[nm], dtype=np.float64)

    # --- hydro_update with the axis repair ------------------------------------
    names = ("hydro_accel", "visc_accel", "visc_heat", "density_calc", "num_density_calc",
             "f_un_neighbor", "dust_density_calc")
    res = refp.hydro_update(*args)
    for nm, a in zip(names, res):
        out["hu_" + nm] = np.array(a, dtype=np.float64)

    # --- loop forms (nsc:673-816), global d injected as the driver does --------
    # choose d so that h(m) = (m/m_0)^(1/3) d is comparable to the kNN radius
    d = float(np.median(h) / np.median((mass[ptype == 0.] / ref.m_0) ** (1. / 3.)))
    ref.d = d
    out["loop_d"] = np.float64(d)
    out["loop_density"] = ref.density(pts, mass, ptype, nb)
    out["loop_dust_density"] = ref.dust_density(pts, mass, nb, ptype, h)
    out["loop_num_dens"] = ref.num_dens(mass, pts, mu, nb)
    out["loop_del_pressure"] = ref.del_pressure(pts, mass, ptype, nb, E, gam)
    av = ref.artificial_viscosity(nb, pts, ptype, h, mass, out["loop_density"], vel, T, gam, mu)
    out["loop_av_accel"], out["loop_av_heat"] = av
    out["loop_crossing_time"] = np.float64(ref.crossing_time(nb, vel, h, ptype))
    imp = ref.net_impulse(pts, mass, h, vel, ptype, nb, f_un)
    out["loop_drag_onto"], out["loop_drag_reaction"] = imp
    out["grain_mass"] = np.array(ref.grain_mass(ref.mineral_densities, ref.mrn_constants), dtype=np.float64)
    out["sigma_effective"] = np.array(
        ref.sigma_effective(ref.mineral_densities, ref.mrn_constants, ref.mu_specie), dtype=np.float64)

    # constants the fixtures were generated with
    out["const_k"] = np.float64(ref.k)
    out["const_amu"] = np.float64(ref.amu)
    out["const_m_h"] = np.float64(ref.m_h)
    out["const_m_0"] = np.float64(ref.m_0)
    out["const_dt_0"] = np.float64(ref.dt_0)
    out["mu_specie"] = np.array(mu_specie, dtype=np.float64)
    out["gamma_specie"] = np.array(gamma_specie, dtype=np.float64)
    import scipy
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__,
                                "python " + sys.version.split()[0]])
    path = os.path.join(HERE, c["name"] + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f kB" % (os.path.getsize(path) / 1e3))


def main():
    if not os.path.exists(REF):
        print("reference not present (%s): nothing to do" % REF)
        return 0
    import contextlib
    import io
    ref = load_reference(patch_axis=False)
    refp = load_reference(patch_axis=True)
    np.seterr(all="ignore")
    for spec in CASES:
        sink = io.StringIO()                 # the reference prints section timers
        with contextlib.redirect_stdout(sink):
            run_case(ref, refp, spec)
        for line in sink.getvalue().splitlines():
            if line.startswith("wrote"):
                print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
