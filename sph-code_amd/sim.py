"""Device-resident step loop: the fused hot path of sph/code_running.py:217-491
(search -> dt -> density/pressure/viscosity sums -> leapfrog), nothing leaves HBM between steps.

State in, state out keeps the reference's array conventions (sph/code_running.py:114-177):
points/velocities (N,3), mass, particle_type (f64 0/1/2), f_un (N,S), T, mu_array,
gamma_array, E_internal, total_accel.
"""
import ctypes as C
import logging

import numpy as np

from . import _lib
from ._lib import dp, f64

log = logging.getLogger("sph_code_amd")


class Simulation:
    def __init__(self, state, n_neigh=40, dist=None, device=None, with_species=False, ctx=None,
                 incremental=False, with_drag=False, gravity=None, G=6.67430e-11, clip_grad=False,
                 forms="hydro_update", d=None, gravity_order=2, agb=None):
        """with_species: carry f_un on the device; every step (hydro_update mode) then also forms the species number
        densities F[s,i] of nsc:624-627 on its own neighbour list.  agb = (splines, mapto, divisor) as
        sph_code_amd.agb.interpolate_amounts returns them: the same pass also leaves the per-particle metallicity
        (the expression of code_running.py:663) and the AGB dust yields of config_helper.py:183-189 (download_species)."""
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self.ctx.set_incremental(incremental)
        self.k = int(n_neigh)
        self.dist = 0.0 if dist is None or not np.isfinite(dist) else float(dist)
        self.first = True
        pts = f64(state["points"])
        n = pts.shape[0]
        self.n = n
        vel = f64(state["velocities"], (n, 3))
        fu = f64(state["f_un"]) if (with_species and state.get("f_un") is not None) else None
        acc = state.get("total_accel")
        acc = f64(acc, (n, 3)) if acc is not None else None
        c = self.ctx
        c.check(c.lib.sphx_state_upload(
            c.h, n, 0 if fu is None else fu.shape[1], dp(pts), dp(vel), dp(f64(state["mass"], (n,))),
            dp(f64(state["particle_type"], (n,))), dp(fu), dp(f64(state["T"], (n,))),
            dp(f64(state["mu_array"], (n,))), dp(f64(state["gamma_array"], (n,))),
            dp(f64(state["E_internal"], (n,))), dp(acc)))
        c.check(c.lib.sphx_set_clip_grad(c.h, 1 if clip_grad else 0))
        if forms not in ("hydro_update", "loop"):
            raise ValueError("forms must be 'hydro_update' or 'loop'")
        if forms == "loop":            # the reference's time loop: nsc.density, del_pressure, ... with the global d
            if d is None:
                raise ValueError("forms='loop' needs the driver's global d (code_running.py:67-68)")
            c.check(c.lib.sphx_state_set_loop_forms(c.h, 1, float(d)))
        if gravity is not None:
            if gravity not in ("direct", "tree"):
                raise ValueError("gravity must be None, 'direct' or 'tree'")
            c.check(c.lib.sphx_state_set_gravity(c.h, 1 if gravity == "direct" else 2, float(G)))
            # tree: cells carry their second moments (order 2, ~0.1 % rms force error) or monopoles only (1, ~1 %)
            c.check(c.lib.sphx_set_gravity_order(c.h, int(gravity_order)))
        self.n_species = 0 if fu is None else fu.shape[1]
        if agb is not None:
            if fu is None:
                raise ValueError("agb needs with_species=True and a state with f_un")
            from . import compat
            splines, mapto, divisor = agb
            i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
            ntx = i32([sp.get_knots()[0].size for sp in splines]); nty = i32([sp.get_knots()[1].size for sp in splines])
            cat = lambda parts: np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.float64).ravel() for q in parts]))
            tx = cat([sp.get_knots()[0] for sp in splines]); ty = cat([sp.get_knots()[1] for sp in splines])
            cf = cat([sp.get_coeffs() for sp in splines]); mp = i32(mapto)
            mu = np.ascontiguousarray(compat.mu_specie[:self.n_species], dtype=np.float64)
            p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
            c.check(c.lib.sphx_state_set_agb(c.h, len(splines), p32(ntx), p32(nty), dp(tx), dp(ty), dp(cf), p32(mp),
                                             float(divisor), dp(mu), compat.solar_mass))
        self.has_agb = agb is not None
        if with_drag:
            # per-particle mean grain mass / cross-section as nsc.net_impulse forms them (nsc:720-726)
            from . import compat
            fu_all = f64(state["f_un"])
            mgm = np.ascontiguousarray(np.sum(compat.grain_mass() * fu_all, axis=1))
            mcs = np.ascontiguousarray(np.sum(compat.sigma_effective() * fu_all, axis=1))
            c.check(c.lib.sphx_state_set_drag(c.h, dp(mgm), dp(mcs)))

    def step(self, nsteps=1, fixed_dt=0.0):
        c = self.ctx
        c.check(c.lib.sphx_step(c.h, int(nsteps), self.k, self.dist, 1 if self.first else 0, float(fixed_dt)))
        self.first = False
        self._log_failures()

    FAILURE_COUNTERS = ("bad_accel", "bad_energy", "bad_state", "bad_h", "short_rows")

    def failures(self):
        """-> dict of the library's failure counters (include/sphx.h sphx_stats: particles whose acceleration / energy /
        updated state / kNN radius was NaN, inf or 0 before the reference's nan_to_num guards, drv:233-238,460-463,490;
        searches that gave up short), summed since the last reset_stats().  All zero in a sane run."""
        st = self.ctx.stats()
        return {k_: int(st[k_]) for k_ in self.FAILURE_COUNTERS}

    def _log_failures(self):
        f = self.failures()
        seen = getattr(self, "_failures_seen", None) or dict.fromkeys(self.FAILURE_COUNTERS, 0)
        new = {k_: f[k_] - seen.get(k_, 0) for k_ in f if f[k_] > seen.get(k_, 0)}
        self._failures_seen = f
        if new:
            log.warning("sphx step: non-finite values met (particles): %s  [cumulative: %s]", new, f)

    def download(self):
        n = self.n
        out = dict(points=np.empty((n, 3)), velocities=np.empty((n, 3)), total_accel=np.empty((n, 3)),
                   E_internal=np.empty(n), T=np.empty(n), sizes=np.empty(n), densities=np.empty(n),
                   num_densities=np.empty(n), visc_heat=np.empty(n))
        dt = C.c_double(0.0)
        c = self.ctx
        c.check(c.lib.sphx_state_download(
            c.h, dp(out["points"]), dp(out["velocities"]), dp(out["total_accel"]), dp(out["E_internal"]),
            dp(out["T"]), dp(out["sizes"]), dp(out["densities"]), dp(out["num_densities"]),
            dp(out["visc_heat"]), C.cast(C.byref(dt), _lib.c_double_p)))
        out["dt"] = dt.value
        # P_i = n_i k_B T_i with n and T of the instant the last step's sums were formed at (the reference forms it and
        # drops it, nsc:608; out["T"] is the temperature AFTER the update, drv:491 - not the one n_i belongs to)
        out["pressure"] = np.empty(n)
        c.check(c.lib.sphx_state_download_pressure(c.h, dp(out["pressure"])))
        return out

    def download_species(self):
        """-> dict: f_un_neighbor (S,N) as nsc.hydro_update returns it (nsc:671), and with an AGB table also
        metallicity (N,) and agb_dust (N,S)."""
        n, S = self.n, self.n_species
        if S < 1:
            raise RuntimeError("the simulation carries no composition (with_species=True and a state with f_un)")
        out = dict(f_un_neighbor=np.empty((S, n)))
        Z = A = None
        if self.has_agb:
            out["metallicity"] = Z = np.empty(n)
            out["agb_dust"] = A = np.empty((n, S))
        c = self.ctx
        c.check(c.lib.sphx_state_download_species(c.h, dp(out["f_un_neighbor"]), dp(Z), dp(A)))
        return out

    # ---- snapshot / restart and per-step diagnostics (the reference only wrote summary
    # statistics at the end of a run, sph/code_running.py:670, and printed the mass-weighted net
    # accelerations every step, sph/code_running.py:465-468) ----------------------------------------
    def snapshot(self, path, state):
        """Write a restartable particle-state snapshot (.npz): the downloaded dynamic arrays plus the
        static per-particle arrays of `state` (mass, particle_type, mu_array, gamma_array, f_un)."""
        d = self.download()
        out = dict(points=d["points"], velocities=d["velocities"], total_accel=d["total_accel"],
                   E_internal=d["E_internal"], T=d["T"], sizes=d["sizes"], dt=np.float64(d["dt"]),
                   first=np.int64(1 if self.first else 0), n_neigh=np.int64(self.k), dist=np.float64(self.dist))
        for key in ("mass", "particle_type", "mu_array", "gamma_array"):
            out[key] = np.asarray(state[key], dtype=np.float64)
        if state.get("f_un") is not None:
            out["f_un"] = np.asarray(state["f_un"], dtype=np.float64)
        np.savez(path, **out)

    @classmethod
    def from_snapshot(cls, path, device=None, **kw):
        z = dict(np.load(path, allow_pickle=False))
        state = {k_: z[k_] for k_ in ("points", "velocities", "total_accel", "E_internal", "T", "mass",
                                       "particle_type", "mu_array", "gamma_array")}
        state["f_un"] = z.get("f_un")
        sim = cls(state, n_neigh=int(z["n_neigh"]), dist=float(z["dist"]), device=device, **kw)
        sim.first = bool(int(z["first"]))
        return sim, state

    def diagnostics(self, state):
        """Mass-weighted net acceleration, momentum and energies of the current state
        (sph/code_running.py:465-468 prints the first of these every step)."""
        d = self.download()
        m = np.asarray(state["mass"], dtype=np.float64)
        mt = m.sum()
        return dict(net_accel=(d["total_accel"] * m[:, None]).sum(axis=0) / mt,
                    momentum=(d["velocities"] * m[:, None]).sum(axis=0),
                    kinetic=0.5 * float((m * (d["velocities"] ** 2).sum(axis=1)).sum()),
                    internal=float(d["E_internal"].sum()), dt=d["dt"])

    def stats(self):
        return self.ctx.stats()

    def reset_stats(self):
        self.ctx.reset_stats()
        self._failures_seen = None
