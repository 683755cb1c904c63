"""The N > 1 path on CPU: recursive-bisection decomposition, 4-phase point-to-point halo
exchange, global dt and migration (sph_code_amd/multigpu.py) run with world_size 2 over gloo.
Compute is played by the CPU oracle (the checker), so what is tested here is the host logic:
the decomposed step must reproduce the single-domain oracle step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

K = 16


class OracleBackend:
    """Test-only stand-in for multigpu.LibBackend built on the oracle's phase hooks."""

    def __init__(self, k):
        from oracle import sph_oracle as orc
        self.orc, self.k = orc, k

    def clamp(self, pos, vel):
        p, v = self.orc.clamp_state(pos.numpy(), vel.numpy())
        pos.copy_(torch.from_numpy(p)); vel.copy_(torch.from_numpy(v))

    def search(self, pos, n_owned, hint, mean_h):
        nb, _, _, _, h = self.orc.neighbors(pos.numpy(), np.inf, self.k, eps=0.0)
        self.nb, self.no = nb, n_owned
        return torch.from_numpy(h.copy())

    def prep(self, pos, vel, m, h, T, mu, gam, ptype):
        self.a = [t.numpy() for t in (pos, vel, m, h, T, mu, gam, ptype)]

    def _hu(self, **kw):
        pos, vel, m, h, T, mu, gam, ptype = self.a
        return self.orc.hydro_update(self.nb, pos, m, h, np.ones((len(pos), 1)), ptype, T, mu, gam, vel, **kw)

    def density(self, want_dust=False):
        out = self._hu()
        self.rhod = torch.from_numpy(out[6].copy()) if want_dust else None
        return torch.from_numpy(out[3].copy()), torch.from_numpy(out[4].copy()), torch.from_numpy(out[0].copy())

    # ---- drag (nsc:719-742) with the mean grain mass / cross-section given per particle; ghost QUERIES contribute
    # nothing (their sums belong to their owners), ghost NEIGHBOURS collect reaction (sent back by the driver) ----
    def drag(self, m, ptype, mgm, mcs):
        o = self.orc
        pos, vel, m_, h, T, mu, gam, pt = self.a
        n = len(pos)
        nb = self.nb.copy()
        nb[self.no:] = np.arange(self.no, n)[:, None]
        pts, nb, dx, r2 = o._loop_geometry(pos, nb)
        dv = vel[nb] - vel[:, None, :]
        mg, mc = mgm.numpy(), mcs.numpy()
        with np.errstate(all="ignore"):
            wf = o._weigh2_dust(r2, m_[nb], h[nb])
            coef = wf / mg[nb] * mc[nb] * np.sum(dv ** 2, axis=2) ** 0.5 * (pt[nb] == 2) * (wf > 0)
            acc = coef[..., None] * dv
        onto = np.sum(acc, axis=1)
        react = np.zeros((n, 3))
        notself = (nb != np.arange(n)[:, None])
        np.add.at(react, nb.ravel(), (-(acc * notself[..., None])).reshape(-1, 3))
        return torch.from_numpy(onto), torch.from_numpy(react)

    def set_drag_terms(self, onto, react, rho, rhod):
        self.terms = (onto.numpy().copy(), react.numpy().copy(), rho.numpy().copy(), rhod.numpy().copy())

    def species(self, fun, m):
        pos, vel, m_, h, T, mu, gam, pt = self.a
        F = self.orc.hydro_update(self.nb, pos, m_, h, fun.numpy(), pt, T, mu, gam, vel)[5]
        return torch.from_numpy(np.ascontiguousarray(F)), None, None

    def pi(self, rho_complete):
        self.rho_c = rho_complete.numpy()
        out, inter = self._hu(rho_in=self.rho_c, return_intermediates=True)
        pos, vel, m, h, T, mu, gam, ptype = self.a
        pt = ptype.copy(); pt[self.no:] = 1.0                    # ghosts do not vote on dt
        ct = self.orc.crossing_time(self.nb, vel, h, pt)
        from sph_code_amd.multigpu import HUGE_CT
        ctv = HUGE_CT if ct == self.orc.DT_0 / 10. else ct - 0.0001
        return torch.from_numpy(inter["Bw"].copy()), ctv

    def visc(self, bw_complete, m):
        out = self._hu(rho_in=self.rho_c, Bj_in=bw_complete.numpy())
        return torch.from_numpy(out[1].copy()), torch.from_numpy(out[2].copy())

    # ---- the loop forms (drv:451-458) ----
    def loop_prep(self, pos, vel, m, T, mu, gam, ptype, E, d):
        self.la = [t.numpy() for t in (pos, vel, m, T, mu, gam, ptype, E)]
        self.d = d

    def loop_pass1(self, h_complete):
        o = self.orc
        pos, vel, m, T, mu, gam, ptype, E = self.la
        self.h = h_complete.numpy()
        rho = o.density(pos, m, ptype, self.nb, self.d)
        rhod = o.dust_density(pos, m, self.nb, ptype, self.h)
        nden = o.num_dens(m, pos, mu, self.nb, self.d)
        delp = o.del_pressure(pos, m, ptype, self.nb, E, gam, self.d)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return f(rho), f(rhod), f(nden), f(delp)

    def loop_pass2(self, rho_complete):
        o = self.orc
        pos, vel, m, T, mu, gam, ptype, E = self.la
        av = o.artificial_viscosity(self.nb, pos, ptype, self.h, m, rho_complete.numpy(), vel, T, gam, mu, self.d)
        pt = ptype.copy(); pt[self.no:] = 1.0                    # ghosts do not vote on dt
        ct = o.crossing_time(self.nb, vel, self.h, pt)
        from sph_code_amd.multigpu import HUGE_CT
        ctv = HUGE_CT if ct == o.DT_0 / 10. else ct - 0.0001
        return torch.from_numpy(np.ascontiguousarray(av[0])), torch.from_numpy(np.ascontiguousarray(av[1])), ctv

    def integrate_loop(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, delp, rho, va, vh, dt, red2=None, first=False,
                       fixed_dt=0.0):
        o = self.orc
        n = lambda t: t.numpy()[:no]
        pa, visc = o.assemble_loop(n(delp), n(rho), None, n(ptype), n(va), None)
        p, v, tot, En, Tn = o.leapfrog(n(pos), n(vel), n(acc), n(E), n(m), n(mu), n(gam), pa, visc, n(vh), dt)
        for dst, src in ((pos, p), (vel, v), (acc, tot), (E, En), (T, Tn)):
            dst[:no] = torch.from_numpy(np.ascontiguousarray(src))

    def integrate(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, ha, va, vh, dt):
        n = lambda t: t.numpy()[:no]
        drag = getattr(self, "terms", None)
        self.terms = None
        p, v, tot, En, Tn = self.orc.integrate(n(pos), n(vel), n(acc), n(E), n(m), n(mu), n(gam), n(ptype),
                                               n(ha), n(va), n(vh), dt, None, drag)
        for dst, src in ((pos, p), (vel, v), (acc, tot), (E, En), (T, Tn)):
            dst[:no] = torch.from_numpy(np.ascontiguousarray(src))


class OracleFusedBackend(OracleBackend):
    """OracleBackend + the tensor-library forms of the library's fused helpers (sphx_dev_reach_dt, sphx_dev_step_scalars,
    sphx_dev_integrate_auto / _loop with the verdict and dt taken from device memory): with them DistributedSim takes the path
    it takes on the GPU - the update launched before the host reads the step's scalars, the NEXT step's plan made at the end
    of the current one - over gloo on the CPU."""

    cap = 0.0

    def set_reach_cap(self, cap_abs):
        self.cap = float(cap_abs)

    def _reach(self, h, vel, halo, skin, dtv):
        speed = torch.sqrt((vel * vel).sum(dim=1))
        if self.cap > 0.0:
            cap = torch.full_like(h, self.cap)
            a = torch.minimum((halo + skin - 1.0) * h, cap)
            b = torch.minimum((halo - 1.0) * h, cap)
            return torch.maximum(h + a, (h + b) + speed * dtv)
        return torch.maximum((halo + skin) * h, halo * h + speed * dtv)

    def reach(self, h, vel, halo_scale, skin_frac, dt_last):
        return self._reach(h, vel, halo_scale, skin_frac, float(dt_last))

    def reach_dt(self, h, vel, halo_scale, skin_frac, dt_dev):
        return self._reach(h, vel, halo_scale, skin_frac, float(dt_dev.reshape(-1)[0]))

    def plan_mask(self, pos, g_lo, g_cs, G, maps, rank):
        lo = torch.tensor(g_lo, dtype=torch.float64)
        c = torch.floor((pos - lo[None]) / g_cs).to(torch.int64).clamp_(0, G - 1)
        cell = (c[:, 2] * G + c[:, 1]) * G + c[:, 0]
        mask = maps[:, cell] != 0
        mask[rank] = False
        return mask, mask.sum(dim=1)

    def step_scalars(self, n_owned, h, w_plan, D, hclip, ct):
        ho = h[:n_owned]
        bad = (ho + 2.0 * D > w_plan).any().to(torch.float64).reshape(1)
        keep = (ho <= hclip) if hclip > 0.0 else torch.ones_like(ho, dtype=torch.bool)
        hm = (ho * keep).sum() / keep.sum().clamp(min=1)
        return torch.cat([bad, -ct.reshape(-1)[:1].to(torch.float64), ho.max().reshape(1), hm.reshape(1)])

    def _dt(self, red2, first, fixed_dt):
        o = self.orc
        ct_min = -float(red2[1])
        ctv = o.DT_0 / 10. if ct_min >= float("inf") else ct_min + 0.0001
        return fixed_dt if fixed_dt > 0 else o.timestep(ctv, first)

    def integrate_auto(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, ha, va, vh, red2, first, fixed_dt):
        if float(red2[0]) > 0.5:                        # the halo was too thin: the state stays as it is
            return torch.zeros(1, dtype=torch.float64)
        dt = self._dt(red2, first, fixed_dt)
        self.integrate(no, pos, vel, acc, E, T, m, mu, gam, ptype, ha, va, vh, dt)
        return torch.tensor([dt], dtype=torch.float64)

    def integrate_loop(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, delp, rho, va, vh, dt, red2=None, first=False,
                       fixed_dt=0.0):
        if red2 is None:
            return OracleBackend.integrate_loop(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, delp, rho, va, vh, dt)
        if float(red2[0]) > 0.5:
            return torch.zeros(1, dtype=torch.float64)
        dt = self._dt(red2, first, fixed_dt)
        OracleBackend.integrate_loop(self, no, pos, vel, acc, E, T, m, mu, gam, ptype, delp, rho, va, vh, dt)
        return torch.tensor([dt], dtype=torch.float64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, nsteps, workload, out_dir, sim_kw=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    sim_kw = dict(sim_kw or {})
    full = bool(sim_kw.get("with_drag") or sim_kw.get("with_species"))
    state = ics.WORKLOADS[workload](n) if full else ics.WORKLOADS[workload](n, light=True)
    mine, lo, hi = mg.decompose_state(state, world, rank)
    if sim_kw.get("forms") == "loop":
        sim_kw["d"] = ics.loop_d(state, K)
    fused = sim_kw.pop("_fused_backend", False)
    force = sim_kw.pop("_force_replan", 0)
    sim = mg.DistributedSim(mine, lo, hi, (OracleFusedBackend if fused else OracleBackend)(K), rank, world, device="cpu", **sim_kw)
    if force:
        sim.force_replan = force                     # the regime of the reference's dt rule: a new plan every step
    for _ in range(nsteps):
        sim.step()
    res = sim.owned_numpy()
    if "f_un_neighbor" in res:
        res["f_un_neighbor"] = np.ascontiguousarray(res["f_un_neighbor"].T)      # (n_owned, S): rows concatenate by rank
    res["stats"] = np.array([sim.stats["ghosts"], sim.stats["redo"], sim.stats["migrated"], sim.ex.bytes_sent,
                             sim.stats.get("replans", 0)])
    res["ahead_ms"] = np.float64(sim.host_ms.get("plan_ahead", 0.0))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, n, nsteps, workload, tmp_path, sim_kw=None):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nsteps, workload, str(tmp_path), sim_kw), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(n)), "particles lost or duplicated"
    order = np.argsort(gid)
    merged = {k_: np.concatenate([p[k_] for p in parts])[order]
              for k_ in ("points", "velocities", "total_accel", "E_internal", "T", "sizes", "densities", "f_un_neighbor")
              if k_ in parts[0]}
    merged["dt"] = [float(p["dt"]) for p in parts]
    merged["stats"] = np.sum([p["stats"] for p in parts], axis=0)
    merged["ahead_ms"] = [float(p["ahead_ms"]) for p in parts] if "ahead_ms" in parts[0] else None
    return merged


def _reference(n, nsteps, workload):
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    s = ics.WORKLOADS[workload](n, light=True)
    for it in range(nsteps):
        s = orc.step(s, n_neigh=K, eps=0.0, first=(it == 0))
    return s


def test_rcb_regions_balanced_and_tiling():
    from sph_code_amd import multigpu as mg
    rs = np.random.RandomState(3)
    pts = rs.normal(size=(5000, 3)) * np.array([3., 1., 2.])
    for world in (1, 2, 3, 4, 8):
        owner, lo, hi = mg.rcb_regions(pts, world)
        cnt = np.bincount(owner, minlength=world)
        assert cnt.sum() == 5000 and cnt.max() - cnt.min() <= world
        back = mg.region_of(torch.from_numpy(pts), torch.from_numpy(lo), torch.from_numpy(hi)).numpy()
        assert np.array_equal(back, owner)
        probe = rs.normal(size=(2000, 3)) * 50.
        inside = ((probe[:, None, :] >= lo[None]) & (probe[:, None, :] <= hi[None])).all(axis=2)
        assert (inside.sum(axis=1) >= 1).all()                    # regions tile space
        d = mg.dist_to_region(torch.from_numpy(pts), torch.from_numpy(lo[0]), torch.from_numpy(hi[0])).numpy()
        assert (d[owner == 0] == 0).all() and (d >= 0).all()


def test_world1_matches_oracle_step():
    """DistributedSim with one rank (no communication) == the oracle's single-domain step."""
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    n, nsteps = 1500, 3
    state = ics.uniform_sphere(n, light=True)
    mine, lo, hi = mg.decompose_state(state, 1, 0)
    sim = mg.DistributedSim(mine, lo, hi, OracleBackend(K), 0, 1, device="cpu")
    for _ in range(nsteps):
        sim.step()
    got = sim.owned_numpy()
    ref = _reference(n, nsteps, "uniform_sphere")
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-13)
    np.testing.assert_allclose(got["velocities"], ref["velocities"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)
    assert got["dt"] == pytest.approx(ref["dt"], rel=1e-14)


@pytest.mark.parametrize("workload", ["uniform_sphere", "polytrope"])
def test_world2_gloo_matches_single_domain(workload, tmp_path):
    """world_size 2 over gloo: halo exchange + global dt + migration reproduce the single-domain
    result (same neighbour sets in the same order -> same sums)."""
    n, nsteps = 3000, 3
    got = _run_world(2, n, nsteps, workload, tmp_path)
    ref = _reference(n, nsteps, workload)
    assert got["dt"][0] == got["dt"][1] == pytest.approx(ref["dt"], rel=1e-14)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-13)
    np.testing.assert_allclose(got["velocities"], ref["velocities"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got["E_internal"], ref["E_internal"], rtol=1e-12)
    ghosts, redo, migrated, sent, replans = got["stats"]
    assert ghosts > 0 and sent > 0                                # the halo really was exchanged
    if workload == "polytrope":          # slow drift relative to h: the send lists are reused
        assert replans < 2 * nsteps      # (summed over the 2 ranks)


def test_world4_gloo_runs(tmp_path):
    n, nsteps = 2400, 2
    got = _run_world(4, n, nsteps, "uniform_sphere", tmp_path)
    ref = _reference(n, nsteps, "uniform_sphere")
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-13)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)


def test_world8_gloo_runs(tmp_path):
    """The rank count of the node's scaling run: 2 x 2 x 2 regions, every rank with 7 possible peers."""
    n, nsteps = 4000, 2
    got = _run_world(8, n, nsteps, "polytrope", tmp_path)
    ref = _reference(n, nsteps, "polytrope")
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-13)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-13)
    assert len(set(got["dt"])) == 1


def test_world2_eight_steps_delayed_migration_and_forced_replans(tmp_path):
    """Eight steps under the reference's large dt: the plan goes stale every step (after three in a
    row the staleness reductions are skipped and every step replans), strays are handed over only on
    every 4th replan - the result is still the single-domain one."""
    n, nsteps = 2000, 8
    got = _run_world(2, n, nsteps, "uniform_sphere", tmp_path)
    ref = _reference(n, nsteps, "uniform_sphere")
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-12)
    ghosts, redo, migrated, sent, replans = got["stats"]
    assert ghosts > 0 and replans >= nsteps and migrated > 0          # (summed over the ranks)


@pytest.mark.parametrize("forms", ["hydro_update", "loop"])
def test_world4_plan_made_ahead_with_redo_and_migration(tmp_path, forms):
    """The path the GPU driver takes by default, on four ranks over gloo (ADVICE, round 2: it had only run with two ranks
    sharing one GPU): the update launched before the host reads the step's scalars (verdict and dt 'on the device'), the
    NEXT step's plan - reach, need map, both all_gathers, send mask - made at the end of the current one while deferred
    exchanges may still be in flight, every step a replan (force_replan), no head-room in the claims so that some steps are
    REDONE (a redo drops the plan made ahead), particles migrating on every 2nd replan inside that window.  The result
    is still the single-domain oracle's."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps = 3000, 6
    wl = "uniform_cube" if forms == "loop" else "polytrope"      # (quiet for the six steps: no runaway particle to compare)
    kw = dict(_fused_backend=True, _force_replan=64, halo_scale=1.0, skin_frac=0.0, migrate_every=2)
    if forms == "loop":
        kw["forms"] = "loop"
    got = _run_world(4, n, nsteps, wl, tmp_path, sim_kw=kw)
    ghosts, redo, migrated, sent, replans = got["stats"]
    assert redo > 0 and migrated > 0 and replans >= 4 * nsteps
    assert all(t > 0.0 for t in got["ahead_ms"])                    # every rank did make plans ahead
    if forms == "loop":
        ref = ics.uniform_cube(n, light=True)
        d = ics.loop_d(ref, K)
        for it in range(nsteps):
            ref = orc.step_loop(ref, d, n_neigh=K, eps=0.0, first=(it == 0))
    else:
        ref = _reference(n, nsteps, wl)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-12)
    assert len(set(got["dt"])) == 1 and got["dt"][0] == pytest.approx(ref["dt"], rel=1e-14)


def test_world2_halo_too_thin_is_detected_and_redone(tmp_path):
    """No head-room in the claimed reach (halo_scale 1, no skin): after the first move some kNN radius
    outgrows its claim, which the end-of-step verdict (one reduction with dt) must catch - the step's
    search and sums are redone with a wider halo and the result is still the single-domain one."""
    n, nsteps = 2000, 3
    got = _run_world(2, n, nsteps, "uniform_sphere", tmp_path, sim_kw=dict(halo_scale=1.0, skin_frac=0.0))
    ref = _reference(n, nsteps, "uniform_sphere")
    ghosts, redo, migrated, sent, replans = got["stats"]
    assert redo > 0
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-12)
    assert got["dt"][0] == got["dt"][1] == pytest.approx(ref["dt"], rel=1e-14)


def test_world2_gloo_loop_forms_on_the_reference_ic(tmp_path):
    """BASELINE configs[3] in small: the reference's own IC (uniform cube, drv:62,132) stepped by the loop forms its
    time loop calls (drv:451-458), two ranks over gloo - three halo phases (state + E, h_j, rho_j) - against the
    single-domain oracle.step_loop (whose loop forms the golden vectors pin)."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps = 3000, 4
    got = _run_world(2, n, nsteps, "uniform_cube", tmp_path, sim_kw=dict(forms="loop"))
    ref = ics.uniform_cube(n, light=True)
    d = ics.loop_d(ref, K)
    ref["f_un"] = None
    for it in range(nsteps):
        ref = orc.step_loop(ref, d, n_neigh=K, eps=0.0, first=(it == 0))
    assert got["dt"][0] == got["dt"][1] == pytest.approx(ref["dt"], rel=1e-14)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-13)
    np.testing.assert_allclose(got["velocities"], ref["velocities"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(got["E_internal"], ref["E_internal"], rtol=1e-12)
    assert got["stats"][0] > 0


def _snap_worker(rank, world, port, n, out_dir, dusty=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    state = ics.dusty_sphere(n) if dusty else ics.polytrope_sphere(n, light=True)
    mine, lo, hi = mg.decompose_state(state, world, rank)
    kw = dict(with_drag=True, with_species=True) if dusty else {}
    sim = mg.DistributedSim(mine, lo, hi, OracleBackend(K), rank, world, device="cpu", **kw)
    sim.step(); sim.step()
    prefix = os.path.join(out_dir, "snap")
    sim.snapshot(prefix)
    diag = sim.diagnostics()
    sim.step(); sim.step()
    a = sim.owned_numpy()
    sim2 = mg.DistributedSim.from_snapshot(prefix, OracleBackend(K), rank, world, device="cpu")     # (no kwargs: the file's flags)
    assert not sim2.first and sim2.stats["steps"] == 2
    assert sim2.with_drag == dusty and sim2.with_species == dusty
    assert sorted(sim2.extra_fields) == (["fun", "mcs", "mgm"] if dusty else [])
    sim2.step(); sim2.step()
    b = sim2.owned_numpy()
    for r_ in (a, b):
        if "f_un_neighbor" in r_:
            r_["f_un_neighbor"] = np.ascontiguousarray(r_["f_un_neighbor"].T)      # (n_owned, S): rows concatenate by rank
        for k_ in [k_ for k_, v in r_.items() if v is None]:
            del r_[k_]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), diag_net=diag["net_accel"], diag_n=diag["particles"],
             diag_kin=diag["kinetic"], **{"a_" + k_: v for k_, v in a.items()}, **{"b_" + k_: v for k_, v in b.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dusty", [False, True])
def test_world2_snapshot_restart_and_diagnostics(tmp_path, dusty):
    """A decomposed run resumed from its per-rank snapshots continues like the uninterrupted one (the plan is rebuilt:
    same neighbour sets, same sums); the diagnostics are global sums (every rank reports the same numbers).
    dusty: a two-phase run (BASELINE configs[4]'s ingredients: drag with its reverse halo + the species pass) - the
    snapshot carries the composition rows, the drag coefficients and the two flags, and the resumed run, given NO
    keyword, keeps doing both (it used to come back with drag and species silently off)."""
    n, world = 2400, 2
    port = _free_port()
    mp.spawn(_snap_worker, args=(world, port, n, str(tmp_path), dusty), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    assert int(parts[0]["diag_n"]) == int(parts[1]["diag_n"]) == n
    assert np.array_equal(parts[0]["diag_net"], parts[1]["diag_net"]) and float(parts[0]["diag_kin"]) > 0
    if dusty:           # the species pass ran in the resumed run too, with the same result
        fa = np.concatenate([p["a_f_un_neighbor"] for p in parts])[np.argsort(np.concatenate([p["a_gid"] for p in parts]))]
        fb = np.concatenate([p["b_f_un_neighbor"] for p in parts])[np.argsort(np.concatenate([p["b_gid"] for p in parts]))]
        np.testing.assert_allclose(fb, fa, rtol=1e-12, atol=0)
    for key in ("points", "velocities", "sizes", "densities", "E_internal"):
        a = np.concatenate([p["a_" + key] for p in parts])[np.argsort(np.concatenate([p["a_gid"] for p in parts]))]
        b = np.concatenate([p["b_" + key] for p in parts])[np.argsort(np.concatenate([p["b_gid"] for p in parts]))]
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-9 if key == "velocities" else 0, err_msg=key)


def test_world2_gloo_drag_with_reverse_halo_and_species(tmp_path):
    """A dusty sphere over two ranks with the gas-dust drag in the step (nsc:719-742, drv:455,462-463,473): ghosts carry their
    drag coefficients, the reaction scatter-added onto ghost neighbours (nsc:741) travels back to the owners - the reverse
    halo - and the result is the single-domain oracle step's; the species pass (nsc:624-627) of the last step likewise."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps = 3000, 3
    got = _run_world(2, n, nsteps, "dusty_sphere", tmp_path, sim_kw=dict(with_drag=True, with_species=True))
    ref = ics.dusty_sphere(n)
    prev = None
    for it in range(nsteps):
        prev = ref
        ref = orc.step(ref, n_neigh=K, eps=0.0, first=(it == 0), with_drag=True)
    assert got["dt"][0] == got["dt"][1] == pytest.approx(ref["dt"], rel=1e-14)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-14)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-13)
    np.testing.assert_allclose(got["points"], ref["points"], rtol=1e-12)
    np.testing.assert_allclose(got["velocities"], ref["velocities"], rtol=1e-10, atol=1e-9)
    assert np.abs(ref["velocities"] - orc.step(prev, n_neigh=K, eps=0.0, first=False)["velocities"]).max() > 0   # drag acted
    # species sums of the last step: on the state the step started from
    p, v = orc.clamp_state(prev["points"], prev["velocities"])
    F = orc.hydro_update(ref["neighbor"], p, prev["mass"], ref["sizes"], prev["f_un"], prev["particle_type"], prev["T"],
                         prev["mu_array"], prev["gamma_array"], v)[5]
    np.testing.assert_allclose(got["f_un_neighbor"].T, F, rtol=1e-12, atol=0)
