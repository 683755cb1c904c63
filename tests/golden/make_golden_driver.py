#!/usr/bin/env python3
"""Golden vectors for the integrator and the time-step rule from the reference DRIVER's own statements.

CONTAINER-ONLY TOOL (no-op when /root/reference is absent).  sph/code_running.py is a script that does
not run to completion as committed (undefined names, a units slip), so it cannot be
imported or executed as a whole.  Its integrator and dt control are two straight-line statement
blocks inside the time loop:

    code_running.py:222-238   crossing time -> dt rule, position clamps, nan_to_num
    code_running.py:460-491   acceleration assembly, viscous limiter, leapfrog, energy, temperature

This tool reads the file at run time, takes those two blocks BY LINE RANGE from the in-memory text
(tabs expanded as Python 2 tokenises them, dedented from the `while` body), and `exec`s each block in
a namespace of seeded arrays carrying the names the block reads (`points`, `velocities`, `delp`,
`densities`, `av`, ...).  `nsc.crossing_time` is replaced by a stub that returns the seeded crossing
time: the function itself is pinned by the loop-form fixtures of make_golden.py.  What is stored is
data only: the inputs put into the namespace and the values the block left behind.  No source text.
"""
import copy
import io
import os
import sys
import textwrap
import contextlib

import numpy as np

REF = "/root/reference/sph/code_running.py"
HERE = os.path.dirname(os.path.abspath(__file__))

YEAR = 60. * 60. * 24. * 365.
AU = 1.496e11


def block(first, last):
    lines = open(REF).read().split("\n")[first - 1:last]
    return compile(textwrap.dedent("\n".join(l.expandtabs(8) for l in lines)) + "\n",
                   "code_running.py:%d-%d" % (first, last), "exec")


class NscStub(object):
    """Stands in for the module object the driver calls and assigns into (drv:222, drv:231)."""

    def __init__(self, ct):
        self._ct = ct
        self.dt = None

    def crossing_time(self, neighbor, velocities, sizes, particle_type):
        return self._ct


def constants_from_driver():
    """The driver's own module-level constants (drv:25-57,78): evaluate the assignments we need."""
    ns = {"np": np}
    src = open(REF).read().split("\n")
    want = ("year", "AU", "dt_0", "MAX_AGE", "amu", "m_h", "k", "solar_mass")
    import scipy.constants as constants
    ns["constants"] = constants
    for ln in src[:90]:
        s = ln.strip()
        for w in want:
            if s.startswith(w + " =") or s.startswith(w + "="):
                try:
                    exec(s, ns)
                except Exception:
                    pass
    return {w: ns[w] for w in want if w in ns}


def run_dt_block(code, cst, ct, age, points, velocities):
    ns = dict(np=np, nsc=NscStub(ct), age=age, dt_0=cst["dt_0"], MAX_AGE=cst["MAX_AGE"], AU=cst["AU"],
              year=cst["year"], points=points.copy(), velocities=velocities.copy(), neighbor=None, sizes=None,
              particle_type=None)
    with contextlib.redirect_stdout(io.StringIO()):
        exec(code, ns)
    return ns["dt"], ns["nsc"].dt, ns["points"], ns["velocities"], ns["ct"]


def run_leapfrog_block(code, cst, inp):
    ns = dict(np=np, copy=copy, m_h=cst["m_h"], k=cst["k"])
    ns.update({k_: (v.copy() if isinstance(v, np.ndarray) else v) for k_, v in inp.items()})
    ns["viscous_drag"] = (inp["drag_on_gas"].copy(), inp["drag_reaction"].copy())
    ns["av"] = (inp["av_accel"].copy(), inp["av_heat"].copy())
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        exec(code, ns)
    return {k_: np.asarray(ns[k_]) for k_ in ("points", "velocities", "total_accel", "old_accel", "E_internal", "T",
                                              "pressure_accel", "drag_accel_gas", "drag_accel_dust", "visc_accel",
                                              "dv")}


def leapfrog_inputs(rs, n, cst, with_dust, first_shape_mismatch=False):
    ptype = np.zeros(n)
    if with_dust:
        ptype[rs.rand(n) < 0.15] = 2.0
        ptype[rs.rand(n) < 0.03] = 1.0
    mass = rs.uniform(0.5, 1.5, n) * 0.4 * cst["solar_mass"] / 715.
    mu = rs.uniform(2.0, 2.4, n)
    gam = rs.uniform(1.4, 1.67, n)
    T = rs.uniform(10., 20., n)
    E = gam * mass * cst["k"] * T / (mu * cst["m_h"])
    rho = rs.uniform(0.5, 2.0, n) * 1e-18
    rho[rs.rand(n) < 0.02] = 0.0                       # empty neighbourhoods: 0/0 and x/0 -> nan_to_num
    rho_d = rs.uniform(0.0, 1.0, n) * 1e-20
    vel = rs.normal(0.0, 1000.0, (n, 3))
    slow = rs.rand(n) < 0.2
    vel[slow] *= 1e-6                                   # these trip the viscous limiter (drv:475)
    vel[rs.rand(n) < 0.02] = 0.0                       # exactly at rest: limiter with -0/dt
    inp = dict(
        particle_type=ptype, mass=mass, mu_array=mu, gamma_array=gam, densities=rho, dust_densities=rho_d,
        points=rs.uniform(-1.0, 1.0, (n, 3)) * 0.6e6 * cst["AU"], velocities=vel,
        grav_accel=rs.normal(0.0, 1e-12, (n, 3)),
        delp=rs.normal(0.0, 1e-29, (n, 3)),
        drag_on_gas=rs.normal(0.0, 1e-13, (n, 3)) * with_dust,
        drag_reaction=rs.normal(0.0, 1e-13, (n, 3)) * with_dust,
        av_accel=rs.normal(0.0, 1e-11, (n, 3)),
        av_heat=np.abs(rs.normal(0.0, 1e18, n)),
        total_accel=rs.normal(0.0, 1e-11, (n, 3)),
        E_internal=E,
    )
    inp["av_heat"][rs.rand(n) < 0.02] = np.nan         # nan_to_num on the heating term (drv:490)
    inp["E_internal"][rs.rand(n) < 0.02] = np.nan
    if first_shape_mismatch:                            # drv:484-485: shapes differ -> dv = a dt
        inp["total_accel"] = np.zeros((n + 1, 3))
    return inp


def main():
    if not os.path.exists(REF):
        print("make_golden_driver: /root/reference absent - nothing to do")
        return 0
    cst = constants_from_driver()
    for w in ("year", "AU", "dt_0", "MAX_AGE", "m_h", "k", "solar_mass"):
        assert w in cst, w
    dt_code = block(222, 238)
    lf_code = block(460, 491)
    out = {"const_" + k_: np.float64(v) for k_, v in cst.items()}

    # ---- dt rule + clamps (drv:222-238) ----
    rs = np.random.RandomState(4601)
    dt0, mx = cst["dt_0"], cst["MAX_AGE"]
    cts = [0.0, 1e-4, dt0 / 50., dt0 / 5., dt0 / 5. * (1 + 1e-15), dt0 / 3., dt0, 2. * dt0, 2. * dt0 * 1.0000001,
           10. * dt0, mx, np.nextafter(mx, np.inf), 3. * mx, 1e300, np.finfo(np.float64).max, np.inf, dt0 / 10.]
    ages = [0, 0.0, dt0 / 10., 5e5 * cst["year"]]
    lim = 1e11 * cst["AU"]
    n = 96
    pts = rs.uniform(-1, 1, (n, 3)) * 1e6 * cst["AU"]
    vel = rs.normal(0, 1000., (n, 3))
    pts[3, 0] = 2 * lim; pts[4, 1] = -3 * lim; pts[5, 2] = np.inf; pts[6, 0] = -np.inf
    pts[7, 1] = np.nan; pts[8] = [lim, -lim, np.nextafter(lim, np.inf)]
    vel[9, 0] = np.nan; vel[10, 1] = np.inf; vel[11, 2] = -np.inf; vel[3] = [5., 6., 7.]
    rows = []
    for age in ages:
        for ct in cts:
            dt, nsc_dt, p2, v2, ct2 = run_dt_block(dt_code, cst, ct, age, pts, vel)
            assert dt == nsc_dt and (ct2 == ct or (np.isnan(ct) and np.isnan(ct2)))
            rows.append((float(age), float(ct), float(dt)))
    out["dt_cases"] = np.array(rows)                   # columns: age, ct, dt
    out["clamp_points_in"], out["clamp_velocities_in"] = pts, vel
    out["clamp_points_out"], out["clamp_velocities_out"] = p2, v2

    # ---- assembly + limiter + leapfrog + energy (drv:460-491) ----
    cases = [("gas", 256, False, False, dt0 / 10.), ("dusty", 256, True, False, dt0 / 5.),
             ("dusty_long", 192, True, False, 2. * dt0), ("first_shape", 64, False, True, dt0 / 10.),
             ("maxage", 64, True, False, mx / 100.), ("gas_nograv", 128, False, False, dt0 / 5.)]
    names = []
    for name, n, dust, mismatch, dt in cases:
        inp = leapfrog_inputs(np.random.RandomState(4700 + len(names)), n, cst, dust, mismatch)
        inp["dt"] = float(dt)
        if name.endswith("nograv"):                     # the form the device-pointer API integrates (no gravity term)
            inp["grav_accel"] = np.zeros((n, 3))
        res = run_leapfrog_block(lf_code, cst, inp)
        for k_, v in inp.items():
            out["lf_%s_in_%s" % (name, k_)] = np.asarray(v)
        for k_, v in res.items():
            out["lf_%s_out_%s" % (name, k_)] = v
        names.append(name)
    out["lf_cases"] = np.array(names)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(HERE, "driver_integrator.npz"), **out)
    print("wrote driver_integrator.npz: %d dt cases, %d leapfrog cases" % (len(rows), len(names)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
