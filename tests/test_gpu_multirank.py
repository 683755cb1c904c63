"""GPU tests of the decomposed path (sph_code_amd/multigpu.py + libsphx sphx_dev_*): one rank
against the fused single-GPU step loop, and two ranks sharing the one GPU of the test box
(gloo carries the halo; on the 8-GPU node the same code runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
K = 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_one_rank_device_api_matches_fused_step():
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    from sph_code_amd.sim import Simulation
    n, nsteps = 20000, 4
    state = ics.polytrope_sphere(n, light=True)
    sim = Simulation(state, n_neigh=K)
    mine, lo, hi = mg.decompose_state(state, 1, 0)
    dsim = mg.DistributedSim(mine, lo, hi, mg.LibBackend(0, k=K), 0, 1, device="cuda:0")
    for _ in range(nsteps):
        sim.step(1)
        dsim.step()
    a = sim.download()
    b = dsim.owned_numpy()
    order = np.argsort(b["gid"])
    assert a["dt"] == pytest.approx(b["dt"], rel=1e-15)
    for key in ("points", "velocities", "total_accel", "E_internal", "T", "sizes", "densities"):
        np.testing.assert_allclose(b[key][order], a[key], rtol=1e-13, atol=0, err_msg=key)


def test_one_rank_device_api_matches_fused_step_in_loop_form_mode():
    """forms='loop' (the reference's time loop: nsc.density, del_pressure, artificial_viscosity on the step's list):
    the device-pointer protocol of the multi-GPU driver against the fused single-GPU loop, bit for bit."""
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    from sph_code_amd.sim import Simulation
    n, nsteps = 20000, 4
    state = ics.uniform_cube(n, light=True)
    d = ics.loop_d(state, K)
    sim = Simulation(state, n_neigh=K, forms="loop", d=d)
    mine, lo, hi = mg.decompose_state(state, 1, 0)
    dsim = mg.DistributedSim(mine, lo, hi, mg.LibBackend(0, k=K), 0, 1, device="cuda:0", forms="loop", d=d)
    for _ in range(nsteps):
        sim.step(1)
        dsim.step()
    a = sim.download()
    b = dsim.owned_numpy()
    order = np.argsort(b["gid"])
    assert a["dt"] == pytest.approx(b["dt"], rel=1e-15)
    for key in ("points", "velocities", "total_accel", "E_internal", "T", "sizes", "densities"):
        np.testing.assert_allclose(b[key][order], a[key], rtol=1e-13, atol=0, err_msg=key)


def test_one_rank_device_api_search_with_outlier_levels_is_exact(monkeypatch):
    """The device-pointer search (sphx_dev_search: hints by caller index, radii written by id - knn_kernel<0,2,...>) on a
    core with a halo of escapers spread over five decades: with the outlier levels built (SPHX_OUTLIER_LEVELS=1) every
    radius still equals the exact K-th-neighbour distance (cKDTree, eps = 0), hinted and unhinted."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    monkeypatch.setenv("SPHX_OUTLIER_LEVELS", "1")
    n_core, n_halo = 30000, 5000
    rs = np.random.RandomState(5)
    core = (rs.rand(n_core, 3) - 0.5) * 2e17
    u = rs.normal(size=(n_halo, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    pts = np.ascontiguousarray(np.concatenate([core, u * (10.0 ** rs.uniform(17.2, 21.5, n_halo))[:, None]]))
    _, _, _, _, h_ref = orc.neighbors(pts, np.inf, K, eps=0.0)
    be = mg.LibBackend(0, k=K)
    pos = torch.as_tensor(pts, device="cuda:0")
    n = pts.shape[0]
    h0 = be.search(pos, n, None, 0.0)                                   # no hints
    assert np.array_equal(h0.cpu().numpy(), h_ref)
    for scale in (1.0, 0.3, 4.0):                                       # good, too small and far too large hints
        h1 = be.search(pos, n, (h0 * scale).contiguous(), float(h0.mean()))
        assert np.array_equal(h1.cpu().numpy(), h_ref), scale
    st = be.ctx.stats()
    assert st["outlier_levels"] > 0 and st["short_rows"] == 0
    torch.cuda.synchronize()


def _worker(rank, world, port, n, nsteps, workload, out_dir, forms="hydro_update", c5=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sph_code_amd.ics as ics
    from sph_code_amd import multigpu as mg
    state = ics.WORKLOADS[workload](n) if c5 else ics.WORKLOADS[workload](n, light=True)
    mine, lo, hi = mg.decompose_state(state, world, rank)
    kw = dict(forms="loop", d=ics.loop_d(state, K)) if forms == "loop" else {}
    if c5:        # BASELINE configs[4]'s ingredients: drag with its reverse halo, species pass, metallicity, AGB yields
        import sph_code_amd.agb as agb
        g = np.load(os.path.join(ROOT, "tests", "golden", "agb_reference.npz"))
        kw.update(with_drag=True, with_species=True,
                  agb=agb.splines_from_arrays(g["tx"], g["ty"], g["coeffs"], g["mapto"], float(g["divisor"])))
    sim = mg.DistributedSim(mine, lo, hi, mg.LibBackend(0, k=K), rank, world, device="cuda:0", comm_device="cpu", **kw)
    if os.environ.get("SPHX_TEST_FORCE_REPLAN"):      # the regime of the reference's dt rule: a new plan every step
        sim.force_replan = int(os.environ["SPHX_TEST_FORCE_REPLAN"])
        sim.plan_ahead = os.environ.get("SPHX_TEST_PLAN_AHEAD", "1") != "0"
    if os.environ.get("SPHX_TEST_THIN_PLANS"):        # no head-room at all: any radius that grows makes the step redo
        sim.halo_scale, sim.skin_frac, sim.reach_vfac = 1.0, 0.0, 0.0
        sim.migrate_every = 3                         # (with a redo per step: migrations fall on redos too)
    n_ahead = 0
    for _ in range(nsteps):
        sim.step()
        n_ahead += sim.plan_next is not None
    res = sim.owned_numpy()
    res["plans_made_ahead"] = np.array(n_ahead)
    if "f_un_neighbor" in res:
        res["f_un_neighbor"] = np.ascontiguousarray(res["f_un_neighbor"].T)      # (n_owned, S): rows concatenate by rank
    res["stats"] = np.array([sim.stats["ghosts"], sim.stats["redo"], sim.stats["migrated"], sim.ex.bytes_sent,
                             sim.stats.get("replans", 0)])
    res["blob_split"] = np.array(sim.backend.blob_split_counts())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,workload", [(2, "polytrope"), (4, "uniform_sphere")])
def test_ranks_sharing_one_gpu_match_oracle(world, workload, tmp_path):
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps = 12000, 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nsteps, workload, str(tmp_path)), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    got = {k_: np.concatenate([p[k_] for p in parts])[order]
           for k_ in ("points", "velocities", "E_internal", "sizes", "densities")}
    ref = ics.WORKLOADS[workload](n, light=True)
    for it in range(nsteps):
        ref = orc.step(ref, n_neigh=K, eps=0.0, first=(it == 0))
    assert float(parts[0]["dt"]) == pytest.approx(ref["dt"], rel=1e-12)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-11)
    assert np.max(np.abs(got["points"] - ref["points"])) <= 1e-9 * np.max(np.abs(ref["points"]))
    assert np.max(np.abs(got["velocities"] - ref["velocities"])) <= 1e-9 * np.max(np.abs(ref["velocities"]))
    np.testing.assert_allclose(got["E_internal"], ref["E_internal"], rtol=1e-9)
    assert sum(p["stats"][0] for p in parts) > 0


@pytest.mark.parametrize("forms,workload", [("hydro_update", "polytrope"), ("loop", "uniform_cube")])
def test_two_ranks_match_the_fused_step_at_2e5_particles(forms, workload, tmp_path):
    """The decomposed step at a size where a rank has a thousand workgroups of both kinds (interior ones under the halo
    phases, boundary ones after), ghosts appended in place behind 10^5 owned particles, plans made ahead: against the
    single-GPU fused loop on the same state (itself pinned to the oracle at small sizes and to the array API at 10^6)."""
    import sph_code_amd.ics as ics
    from sph_code_amd.sim import Simulation
    n, nsteps, world = 200000, 4, 2
    mp.spawn(_worker, args=(world, _free_port(), n, nsteps, workload, str(tmp_path), forms), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    assert all(p["blob_split"][0] > 100 and p["blob_split"][1] > 20 for p in parts)
    order = np.argsort(np.concatenate([p["gid"] for p in parts]))
    got = {k_: np.concatenate([p[k_] for p in parts])[order]
           for k_ in ("points", "velocities", "total_accel", "E_internal", "sizes", "densities")}
    state = ics.WORKLOADS[workload](n, light=True)
    kw = dict(forms="loop", d=ics.loop_d(state, K)) if forms == "loop" else {}
    sim = Simulation(state, n_neigh=K, **kw)
    sim.step(nsteps)
    ref = sim.download()
    assert float(parts[0]["dt"]) == pytest.approx(ref["dt"], rel=1e-13)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-13)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-12)
    np.testing.assert_allclose(got["E_internal"], ref["E_internal"], rtol=1e-11)
    for k_ in ("points", "velocities", "total_accel"):
        assert np.max(np.abs(got[k_] - ref[k_])) <= 1e-10 * np.max(np.abs(ref[k_])), k_


def test_two_ranks_redone_steps_match_oracle(tmp_path, monkeypatch):
    """Plans with no head-room (claimed reach = the previous radius itself, no skin, no velocity term): the post-search verdict finds them too thin, the failing
    particles claim more, everybody replans - migrating by the order of the attempt just made when a migration is due -
    and the step's search and sums are redone.  Same trajectories as the oracle's single-domain step."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps, world = 12000, 4, 2
    monkeypatch.setenv("SPHX_TEST_THIN_PLANS", "1")
    mp.spawn(_worker, args=(world, _free_port(), n, nsteps, "polytrope", str(tmp_path)), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    assert min(int(p["stats"][1]) for p in parts) >= 2                   # redone steps (the same count on every rank)
    order = np.argsort(np.concatenate([p["gid"] for p in parts]))
    got = {k_: np.concatenate([p[k_] for p in parts])[order] for k_ in ("points", "velocities", "sizes", "densities")}
    ref = ics.WORKLOADS["polytrope"](n, light=True)
    for it in range(nsteps):
        ref = orc.step(ref, n_neigh=K, eps=0.0, first=(it == 0))
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-11)
    assert np.max(np.abs(got["points"] - ref["points"])) <= 1e-9 * np.max(np.abs(ref["points"]))
    assert np.max(np.abs(got["velocities"] - ref["velocities"])) <= 1e-9 * np.max(np.abs(ref["velocities"]))


@pytest.mark.parametrize("forms", ["hydro_update", "loop"])
def test_two_ranks_interior_blobs_under_the_halo_phases_change_nothing(forms, tmp_path, monkeypatch):
    """The search sorts its workgroups of 128 particles by what they need from other ranks; the sums of the interior
    ones (no ghost among the neighbours of their owned particles) are launched while the scalar halo phase they do not
    depend on is in flight, the boundary ones after it (SPHX_MG_OVERLAP, default on).  Same results, bit for bit, as
    running every workgroup after the exchange."""
    n, nsteps, world = 20000, 5, 2
    res = {}
    for ov in ("1", "0"):
        monkeypatch.setenv("SPHX_MG_OVERLAP", ov)
        out = tmp_path / ("ov" + ov)
        out.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), n, nsteps, "uniform_cube" if forms == "loop" else "polytrope", str(out), forms),
                 nprocs=world, join=True)
        parts = [dict(np.load(os.path.join(str(out), "rank%d.npz" % r))) for r in range(world)]
        order = np.argsort(np.concatenate([p["gid"] for p in parts]))
        res[ov] = {k_: np.concatenate([p[k_] for p in parts])[order]
                   for k_ in ("points", "velocities", "E_internal", "sizes", "densities", "total_accel")}
        res[ov]["dt"] = float(parts[0]["dt"])
        split = np.array([p["blob_split"] for p in parts])
    assert (split[:, 0] > 0).all() and (split[:, 1] > 0).all()        # both kinds of workgroup on both ranks
    assert res["1"]["dt"] == res["0"]["dt"]
    for k_ in ("points", "velocities", "E_internal", "sizes", "densities", "total_accel"):
        assert np.array_equal(res["1"][k_], res["0"][k_]), k_


@pytest.mark.parametrize("forms", ["hydro_update", "loop"])
def test_two_ranks_plan_made_ahead_equals_plan_made_at_step_start(forms, tmp_path, monkeypatch):
    """Under the reference's dt rule every step needs a new halo plan.  The driver then makes the NEXT step's plan at
    the end of the current one - reach from the device-side dt (sphx_dev_reach_dt), need map, both all_gathers and the
    send mask (sphx_dev_plan_mask) launched behind the leapfrog update, the counts folded into the step's one host
    read - except on the steps that migrate particles.  Same trajectories, bit for bit, as planning at step start."""
    import sph_code_amd.ics as ics
    n, nsteps, world = 12000, 7, 2
    res = {}
    for ahead in ("1", "0"):
        monkeypatch.setenv("SPHX_TEST_FORCE_REPLAN", "16")
        monkeypatch.setenv("SPHX_TEST_PLAN_AHEAD", ahead)
        out = tmp_path / ("ahead" + ahead)
        out.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), n, nsteps, "uniform_cube" if forms == "loop" else "polytrope", str(out), forms),
                 nprocs=world, join=True)
        parts = [dict(np.load(os.path.join(str(out), "rank%d.npz" % r))) for r in range(world)]
        order = np.argsort(np.concatenate([p["gid"] for p in parts]))
        res[ahead] = {k_: np.concatenate([p[k_] for p in parts])[order]
                      for k_ in ("points", "velocities", "E_internal", "sizes", "densities", "total_accel")}
        res[ahead]["n_ahead"] = [int(p["plans_made_ahead"]) for p in parts]
        res[ahead]["replans"] = [int(p["stats"][4]) for p in parts]
    assert min(res["1"]["n_ahead"]) >= 3 and max(res["0"]["n_ahead"]) == 0      # (migration steps plan at step start)
    assert res["1"]["replans"] == res["0"]["replans"]
    for k_ in ("points", "velocities", "E_internal", "sizes", "densities", "total_accel"):
        assert np.array_equal(res["1"][k_], res["0"][k_]), k_


def test_two_ranks_sharing_one_gpu_loop_forms_on_the_reference_ic(tmp_path):
    """BASELINE configs[3] in small on the GPU: uniform cube (drv:62,132), loop-form step mode, two ranks (gloo
    carries the three halo phases) against the single-domain oracle.step_loop."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    n, nsteps, world = 12000, 3, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nsteps, "uniform_cube", str(tmp_path), "loop"), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    got = {k_: np.concatenate([p[k_] for p in parts])[order]
           for k_ in ("points", "velocities", "E_internal", "sizes", "densities")}
    ref = ics.uniform_cube(n, light=True)
    d = ics.loop_d(ref, K)
    ref["f_un"] = None
    for it in range(nsteps):
        ref = orc.step_loop(ref, d, n_neigh=K, eps=0.0, first=(it == 0))
    assert float(parts[0]["dt"]) == pytest.approx(ref["dt"], rel=1e-12)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-11)
    assert np.max(np.abs(got["points"] - ref["points"])) <= 1e-9 * np.max(np.abs(ref["points"]))
    assert np.max(np.abs(got["velocities"] - ref["velocities"])) <= 1e-9 * np.max(np.abs(ref["velocities"]))
    np.testing.assert_allclose(got["E_internal"], ref["E_internal"], rtol=1e-9)
    assert sum(p["stats"][0] for p in parts) > 0


@pytest.mark.parametrize("forms", ["hydro_update", "loop"])
def test_two_ranks_sharing_one_gpu_drag_species_agb(forms, tmp_path):
    """BASELINE configs[4]'s ingredients in the decomposed step: two-phase gas + dust (drv:120-152), the gas-dust drag with its
    reverse halo (nsc:741: reaction onto ghost neighbours goes back to their owners), the species pass (nsc:624-627) with the
    metallicity and the AGB yields fused in - two ranks sharing the GPU, against the single-domain oracle, in both step modes."""
    from oracle import sph_oracle as orc
    from oracle import agb_oracle as ao
    import sph_code_amd.ics as ics
    n, nsteps, world = 12000, 3, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nsteps, "two_phase", str(tmp_path), forms, True), nprocs=world, join=True)
    parts = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))) for r in range(world)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(n))
    order = np.argsort(gid)
    got = {k_: np.concatenate([p[k_] for p in parts])[order]
           for k_ in ("points", "velocities", "E_internal", "sizes", "densities", "f_un_neighbor", "metallicity", "agb_dust")}
    ref = ics.two_phase(n)
    d = ics.loop_d(ref, K)
    prev = None
    for it in range(nsteps):
        prev = ref
        ref = (orc.step_loop(ref, d, n_neigh=K, eps=0.0, first=(it == 0), with_drag=True) if forms == "loop"
               else orc.step(ref, n_neigh=K, eps=0.0, first=(it == 0), with_drag=True))
    assert float(parts[0]["dt"]) == pytest.approx(ref["dt"], rel=1e-12)
    np.testing.assert_allclose(got["sizes"], ref["sizes"], rtol=1e-12)
    np.testing.assert_allclose(got["densities"], ref["densities"], rtol=1e-10)
    assert np.max(np.abs(got["points"] - ref["points"])) <= 1e-9 * np.max(np.abs(ref["points"]))
    assert np.max(np.abs(got["velocities"] - ref["velocities"])) <= 1e-7 * np.max(np.abs(ref["velocities"]))
    # the last step's species sums, metallicity and yields (on the state that step started from)
    p, v = orc.clamp_state(prev["points"], prev["velocities"])
    F = orc.hydro_update(ref["neighbor"], p, prev["mass"], ref["sizes"], prev["f_un"], prev["particle_type"], prev["T"],
                         prev["mu_array"], prev["gamma_array"], v)[5]
    np.testing.assert_allclose(got["f_un_neighbor"].T, F, rtol=1e-11, atol=0)
    mu = orc.MU_SPECIE
    with np.errstate(all="ignore"):
        Z = (F[6:] * mu[6:, None]).sum(axis=0) / (F * mu[:, None]).sum(axis=0)
    fin = np.isfinite(Z)
    np.testing.assert_allclose(got["metallicity"][fin], Z[fin], rtol=1e-10, atol=1e-300)
    g = np.load(os.path.join(ROOT, "tests", "golden", "agb_reference.npz"))
    dust, _ = ao.calculate_interpolation(prev["mass"][fin], got["metallicity"][fin], (list(g["tx"]), list(g["ty"]), list(g["coeffs"])),
                                         mu, np.ones((fin.sum(), 15)), mapto=g["mapto"], divisor=float(g["divisor"]))
    np.testing.assert_allclose(got["agb_dust"][fin], dust, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_fused_row_kernels_match_tensor_forms():
    """sphx_dev_pack_rows / sphx_dev_regroup (one launch each) against the tensor-library forms the
    CPU tests run (DistributedSim._pack / _regroup without a fused backend)."""
    import torch
    from sph_code_amd.multigpu import LibBackend, DistributedSim
    be = LibBackend(0, k=8)
    dev = be.device
    g = torch.Generator(device="cpu").manual_seed(3)
    n, nr = 1000, 37
    pos = torch.rand((n, 3), generator=g, dtype=torch.float64).to(dev)
    m = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
    gid = torch.randint(0, 1 << 40, (n,), generator=g, dtype=torch.int64).to(dev)
    idx = torch.randperm(n, generator=g)[:123].to(dev)

    class Plain:                        # no fused helpers: DistributedSim falls back to torch
        pass
    shim = DistributedSim.__new__(DistributedSim)
    shim.backend = Plain()
    fused = DistributedSim.__new__(DistributedSim)
    fused.backend = be
    fields = [pos, m, gid]
    for ix in (idx, None):
        a, b = shim._pack(ix, fields), fused._pack(ix, fields)
        assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    rows = shim._pack(idx[:nr], fields)
    sel = torch.randperm(n, generator=g)[:800].to(dev)
    for s_, ns in ((sel, 800), (None, n)):
        for r_ in (rows, None):
            a = shim._regroup(s_, ns, r_, fields)
            b = fused._regroup(s_, ns, r_, fields)
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)
    torch.cuda.synchronize()


def test_need_map_kernel_matches_tensor_form():
    """sphx_dev_need_map (one launch, no host sync) gives the very map of the tensor-library form
    (DistributedSim._need_map without a fused backend): owned particles inside, on the edge of and
    outside the coarse grid, reaches from 0 to several cells to the whole grid."""
    import torch
    from sph_code_amd.multigpu import LibBackend, DistributedSim
    be = LibBackend(0, k=8)
    dev = be.device
    g = torch.Generator(device="cpu").manual_seed(11)
    n, G = 5000, 24
    pos = (torch.rand((n, 3), generator=g, dtype=torch.float64) * 1.4 - 0.2)      # some outside [0,1]^3
    pos[:8] = torch.tensor([[0.0, 0.5, 1.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [-3.0, 0.2, 0.2],
                            [0.25, 0.25, 0.25], [1.0 / 24, 2.0 / 24, 3.0 / 24], [0.999999, 0.0, 0.3], [2.0, 2.0, 2.0]],
                           dtype=torch.float64)
    w = torch.rand(n, generator=g, dtype=torch.float64) * 0.03
    w[::97] = 0.0                        # claims nothing
    w[5::211] = 0.2                      # several cells
    w[11::97] = 0.45                     # ~11 cells: marked from the target side (the pyramid of per-cell maxima)
    w[13::389] = 0.8                     # ~19 cells
    w[7] = 1.0 / 24                      # exactly one cell edge
    w[3] = 50.0                          # the whole grid

    class Plain:
        pass

    def sim(backend, pos_, w_):
        s_ = DistributedSim.__new__(DistributedSim)
        s_.backend, s_.device, s_.G = backend, dev, G
        s_.g_lo = torch.zeros(3, dtype=torch.float64, device=dev)
        s_.g_lo_host = [0.0, 0.0, 0.0]
        s_.g_cs = 1.0 / G
        s_.s = dict(pos=pos_.to(dev))
        return s_._need_map(w_.to(dev))

    for sl in (slice(None), slice(8, None), slice(0, 3), slice(0, 0), slice(8, 400), slice(11, 12)):
        a, b = sim(Plain(), pos[sl], w[sl]), sim(be, pos[sl], w[sl])
        assert a.dtype == b.dtype == torch.uint8 and a.shape == b.shape
        assert torch.equal(a, b), int((a != b).sum())
    # a NaN reach (a broken radius) claims everything in both forms - it must not drop out, nor mask its cell's other claims
    wn = w[8:400].clone()
    wn[17] = float("nan")
    a, b = sim(Plain(), pos[8:400], wn), sim(be, pos[8:400], wn)
    assert torch.equal(a, b) and int(b.sum()) == G ** 3
    assert 0 < int(sim(be, pos[8:400], w[8:400]).sum()) < G ** 3      # a non-trivial map
    torch.cuda.synchronize()


def test_fused_reach_and_step_scalars_match_tensor_forms():
    """sphx_dev_reach / sphx_dev_step_scalars (one launch each) against the tensor-library expressions of
    DistributedSim._replan / .step that the CPU tests run."""
    import torch
    from sph_code_amd.multigpu import LibBackend
    be = LibBackend(0, k=8)
    dev = be.device
    g = torch.Generator(device="cpu").manual_seed(5)
    n, no = 70000, 65537
    h = (torch.rand(n, generator=g, dtype=torch.float64) + 0.5).to(dev)
    h[17] = 40.0                                      # an escaper's radius: left out of the robust mean
    vel = torch.randn((n, 3), generator=g, dtype=torch.float64).to(dev)
    halo, skin, dt = 1.15, 0.15, 0.37
    w = be.reach(h, vel, halo, skin, dt)
    speed = torch.sqrt((vel * vel).sum(dim=1))
    w_ref = torch.maximum((halo + skin) * h, halo * h + speed * dt)
    assert float(((w - w_ref).abs() / w_ref).max()) <= 4e-16      # (the same expression, to rounding)
    # head-room capped at an absolute length (sphx_dev_set_reach_cap): the escaper claims h + cap (+ its displacement)
    cap = 0.2
    be.set_reach_cap(cap)
    wc = be.reach(h, vel, halo, skin, dt)
    dt_dev = torch.tensor([dt], dtype=torch.float64, device=dev)
    wc_dt = be.reach_dt(h, vel, halo, skin, dt_dev)
    be.set_reach_cap(0.0)
    capt = torch.full_like(h, cap)
    wc_ref = torch.maximum(h + torch.minimum((halo + skin - 1.0) * h, capt), (h + torch.minimum((halo - 1.0) * h, capt)) + speed * dt)
    assert float(((wc - wc_ref).abs() / wc_ref).max()) <= 4e-16 and torch.equal(wc_dt, wc)
    assert float(wc[17]) <= 40.0 + cap + float(speed[17]) * dt + 1e-9 and float(w[17]) >= 52.0 - 1e-9
    assert bool((wc <= w * (1 + 1e-15)).all()) and bool((wc >= h).all())
    ct = torch.tensor([3.25], dtype=torch.float64, device=dev)
    for D, hmean_prev in ((0.0, 1.0), (0.02, 1.0), (0.4, 0.0)):
        hclip = 8.0 * hmean_prev if hmean_prev > 0 else 0.0
        out = be.step_scalars(no, h, w, D, hclip, ct).tolist()
        ho = h[:no]
        bad = float((ho + 2.0 * D > w[:no]).any())
        if hmean_prev > 0:
            keep = ho <= hclip
            hm = float((ho * keep).sum() / keep.sum().clamp(min=1))
        else:
            hm = float(ho.mean())
        assert out[0] == bad and out[1] == -3.25 and out[2] == float(ho.max())
        assert abs(out[3] - hm) <= 1e-13 * hm
    assert be.step_scalars(no, h, w, 0.0, 8.0, ct).tolist()[0] == 0.0 and be.step_scalars(no, h, w, 0.4, 8.0, ct).tolist()[0] == 1.0
    torch.cuda.synchronize()


def test_integrate_auto_takes_verdict_and_dt_on_the_device():
    """sphx_dev_integrate_auto: with the verdict flag set nothing moves (the driver redoes the step); otherwise
    it is sphx_dev_integrate with dt from the reduced crossing time by the driver's rule (drv:222-229)."""
    import torch
    from sph_code_amd.multigpu import LibBackend, DistributedSim, HUGE_CT
    be = LibBackend(0, k=8)
    dev = be.device
    g = torch.Generator(device="cpu").manual_seed(9)
    n = 5000
    r = lambda *s_: torch.rand(s_, generator=g, dtype=torch.float64).to(dev)
    base = dict(pos=r(n, 3) * 1e17, vel=r(n, 3) * 1e3, acc=r(n, 3) * 1e-12, E=r(n) * 1e30 + 1e30, T=r(n) * 10 + 10,
                m=r(n) * 1e30 + 1e30, mu=r(n) + 1, gam=r(n) * 0.3 + 1.3, ptype=(r(n) > 0.8).to(torch.float64) * 2.0)
    ha, va, vh = r(n, 3) * 1e-12, r(n, 3) * 1e-13, r(n) * 1e10
    DT0, MAXAGE = DistributedSim.DT_0, DistributedSim.MAX_AGE

    def rule(ct, first, fixed):
        ctv = DT0 / 10. if ct >= HUGE_CT else ct + 0.0001
        if fixed > 0:
            return fixed
        dt = DT0 / 10. if first else max(DT0 / 5., min(DT0 * 2., ctv))
        return MAXAGE / 100. if ctv > MAXAGE else dt

    def run(auto, bad, ct, first, fixed):
        s_ = {k_: v.clone() for k_, v in base.items()}
        args = (n, s_["pos"], s_["vel"], s_["acc"], s_["E"], s_["T"], s_["m"], s_["mu"], s_["gam"], s_["ptype"], ha, va, vh)
        if auto:
            red2 = torch.tensor([bad, -ct], dtype=torch.float64, device=dev)
            dt = float(be.integrate_auto(*args, red2, first, fixed)[0])
        else:
            dt = rule(ct, first, fixed)
            be.integrate(*args, dt)
        return s_, dt

    for ct, first, fixed in ((DT0 * 0.7, False, 0.0), (DT0 * 0.01, False, 0.0), (DT0 * 50, False, 0.0), (HUGE_CT, False, 0.0),
                             (DT0 * 0.7, True, 0.0), (DT0 * 0.7, False, 12345.0), (MAXAGE * 2, False, 0.0)):
        a, dta = run(True, 0.0, ct, first, fixed)
        b, dtb = run(False, 0.0, ct, first, fixed)
        assert dta == dtb, (ct, first, fixed, dta, dtb)
        for k_ in ("pos", "vel", "acc", "E", "T"):
            assert torch.equal(a[k_], b[k_]), k_
    frozen, dt0 = run(True, 1.0, DT0 * 0.7, False, 0.0)
    assert dt0 == 0.0
    for k_ in ("pos", "vel", "acc", "E", "T"):
        assert torch.equal(frozen[k_], base[k_]), k_
    torch.cuda.synchronize()
