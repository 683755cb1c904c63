"""AGB dust-yield lookup (sph/config_helper.py:138-211).

Golden data: tests/golden/agb_reference.npz - outputs of the reference's own interpolate_amounts /
calculate_interpolation, run in the build container by tests/golden/make_golden_agb.py, together with
the AGB tables (data) they were fitted on.  CPU tests pin the oracle; the GPU test checks the HIP
kernel (through the C ABI) against both.  Tolerance: rtol 1e-13 (fp64, a handful of operations)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-13


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "agb_reference.npz"))


def _splines(g):
    return list(g["tx"]), list(g["ty"]), list(g["coeffs"])


def test_oracle_evaluates_like_the_reference_splines(g):
    from oracle import agb_oracle as ao
    Zg, Mg = np.meshgrid(g["grid_z"], g["grid_m"], indexing="ij")       # includes points outside the table
    for o in range(len(g["coeffs"])):
        v = ao.evaluate(g["tx"][o], g["ty"][o], g["coeffs"][o], Zg.ravel(), Mg.ravel()).reshape(Zg.shape)
        np.testing.assert_allclose(v, g["grid_vals"][o], rtol=RTOL, atol=0)


def test_oracle_calculate_interpolation_vs_reference(g):
    from oracle import agb_oracle as ao
    dust, gas = ao.calculate_interpolation(g["masses"], g["metallicities"], _splines(g), g["mu_specie"],
                                           g["composition"], mapto=g["mapto"], divisor=float(g["divisor"]))
    np.testing.assert_allclose(dust, g["dust_mass_created"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(gas, g["gas_mass_created"], rtol=RTOL, atol=0)
    # reference quirks the restatement must keep: repeated targets, clamped masses, untouched species
    assert (dust[:, [0, 1, 2, 3, 4, 5, 9, 14]] == 0).all()
    kg = g["masses"] > 1e20                     # masses in kg: every lookup sits on the 7 M_sun edge
    edge, _ = ao.calculate_interpolation(np.full(kg.sum(), 7.0), g["metallicities"][kg], _splines(g),
                                         g["mu_specie"], g["composition"][kg])
    assert np.array_equal(dust[kg], edge)


def test_oracle_fit_reproduces_reference_knots(g):
    """SciPy's fit on the stored tables gives the reference's knots and coefficients (single patch)."""
    from oracle import agb_oracle as ao
    tab = g["tables"].copy()
    tab[tab <= 0.] = 1e-30
    tx, ty, cf = ao.fit(tab)
    assert all(len(t) == 4 for t in tx) and all(len(t) == 4 for t in ty)
    np.testing.assert_allclose(np.array(cf), g["coeffs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.array(tx), g["tx"], rtol=0, atol=0)


def test_product_fit_matches_reference(g, tmp_path):
    """sph_code_amd.agb.interpolate_amounts on a directory laid out like the reference's."""
    import sph_code_amd.agb as agb
    d = tmp_path / "agb_interp"
    d.mkdir()
    (tmp_path / "sph").mkdir()
    masses = agb.AGB_masses
    for f, table in zip(g["files"], np.swapaxes(g["tables"], 0, 1)):          # (metallicity, species, mass)
        np.savetxt(d / str(f), np.column_stack([masses, table.T]))
    splines, mapto, div = agb.interpolate_amounts(str(tmp_path / "sph"))
    assert div == 3 and list(mapto) == list(g["mapto"])
    for o, sp in enumerate(splines):
        np.testing.assert_allclose(sp.get_coeffs(), g["coeffs"][o], rtol=1e-12, atol=0)
        np.testing.assert_allclose(sp.get_knots()[0], g["tx"][o])
        np.testing.assert_allclose(sp.get_knots()[1], g["ty"][o])


@pytest.mark.gpu
def test_gpu_yields_vs_reference_and_oracle(g):
    import sph_code_amd.agb as agb
    from oracle import agb_oracle as ao
    splines = [agb.Spline(g["tx"][o], g["ty"][o], g["coeffs"][o]) for o in range(len(g["coeffs"]))]
    dust, gas = agb.calculate_interpolation(g["masses"], g["metallicities"], splines, g["mapto"], float(g["divisor"]),
                                            g["mu_specie"], g["composition"])
    np.testing.assert_allclose(dust, g["dust_mass_created"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(gas, g["gas_mass_created"], rtol=RTOL, atol=0)
    # a true piecewise-bilinear table (s = 0: 8 x 27 knots) on many stars, against the oracle
    tab = g["tables"].copy()
    tab[tab <= 0.] = 1e-30
    fine = agb.fit_tables(tab, s=0)
    assert fine[0].get_knots()[1].size > 20
    rs = np.random.RandomState(5)
    n = 200000
    m = rs.uniform(0.5, 8.0, n); z = rs.uniform(0.0, 0.06, n)
    comp = rs.uniform(0.01, 1.0, (n, 15)); comp /= comp.sum(axis=1)[:, None]
    dust, gas = agb.calculate_interpolation(m, z, fine, g["mapto"], 3, g["mu_specie"], comp)
    spl = ([s.get_knots()[0] for s in fine], [s.get_knots()[1] for s in fine], [s.get_coeffs() for s in fine])
    d0, g0 = ao.calculate_interpolation(m, z, spl, g["mu_specie"], comp, mapto=g["mapto"], divisor=3)
    np.testing.assert_allclose(dust, d0, rtol=RTOL, atol=0)
    np.testing.assert_allclose(gas, g0, rtol=1e-12, atol=0)
    only_dust, none = agb.calculate_interpolation(m[:10], z[:10], fine, g["mapto"], 3, g["mu_specie"])
    assert none is None and np.array_equal(only_dust, dust[:10])


@pytest.mark.gpu
def test_gpu_yields_bad_arguments(g):
    import sph_code_amd.agb as agb
    splines = [agb.Spline(g["tx"][o], g["ty"][o], g["coeffs"][o]) for o in range(len(g["coeffs"]))]
    with pytest.raises(Exception):
        agb.calculate_interpolation([1.0], [0.01], splines, np.full(11, 99), 3, g["mu_specie"])
    with pytest.raises(ValueError):
        agb.calculate_interpolation([1.0, 2.0], [0.01], splines, g["mapto"], 3, g["mu_specie"])


@pytest.mark.gpu
def test_step_species_pass_with_fused_metallicity_and_agb_yields(g, monkeypatch):
    """BASELINE configs[4] in small: two-phase gas + dust (drv:120-152), every step also forms F[s,i] (nsc:624-627) on
    its own neighbour list and, in the same pass, the per-particle metallicity (the expression of drv:663 on the
    smoothed composition) and the AGB dust yields (config_helper.py:183-189) at (Z_i, m_i).  Oracle: a composition of
    pinned pieces - hydro_update's F (golden vectors) -> Z -> agb_oracle.calculate_interpolation (reference outputs)."""
    from oracle import sph_oracle as orc
    from oracle import agb_oracle as ao
    import sph_code_amd.ics as ics
    import sph_code_amd.agb as agb
    from sph_code_amd.sim import Simulation
    n, K = 6000, 40
    s = ics.two_phase(n)
    assert np.isclose((s["particle_type"] == 2).mean(), 0.10, atol=0.002)
    table = agb.splines_from_arrays(g["tx"], g["ty"], g["coeffs"], g["mapto"], float(g["divisor"]))
    sim = Simulation(s, n_neigh=K, with_species=True, agb=table)
    sim.step(1)
    got = sim.download_species()
    st = sim.download()
    # the step's sums on the INITIAL state (clamped, searched) - the oracle on the same
    p, v = orc.clamp_state(s["points"], s["velocities"])
    nb, _, _, _, h = orc.neighbors(p, np.inf, K, eps=0.0)
    np.testing.assert_allclose(st["sizes"], h, rtol=2e-15)
    out = orc.hydro_update(nb, p, s["mass"], h, s["f_un"], s["particle_type"], s["T"], s["mu_array"], s["gamma_array"], v)
    F = out[5]
    np.testing.assert_allclose(got["f_un_neighbor"], F, rtol=1e-13, atol=0)
    mu = orc.MU_SPECIE
    with np.errstate(all="ignore"):
        Z = (F[6:] * mu[6:, None]).sum(axis=0) / (F * mu[:, None]).sum(axis=0)
    fin = np.isfinite(Z)
    assert fin.mean() > 0.99
    np.testing.assert_allclose(got["metallicity"][fin], Z[fin], rtol=1e-12, atol=1e-300)
    assert (np.isnan(got["metallicity"]) == ~fin).all()
    dust, _ = ao.calculate_interpolation(s["mass"][fin], got["metallicity"][fin], _splines(g), mu,
                                         np.ones((fin.sum(), 15)), mapto=g["mapto"], divisor=float(g["divisor"]))
    np.testing.assert_allclose(got["agb_dust"][fin], dust, rtol=1e-12, atol=0)
    assert (got["agb_dust"][fin][:, [7, 8, 10, 11, 12, 13]] > 0).any()
    # The array API's species sums (gathers, list positions added in ascending k) against the step's on the same list.
    # The step's LDS form (sphx_blob.hip blob_species_kernel: each distinct neighbour's composition row staged once per
    # blob) adds four partial sums over k mod 4 as (p0 + p1) + (p2 + p3): equal to rounding; with SPHX_SPECIES_LDS=0 the
    # step runs the gather form too and the two agree bit for bit.
    import sph_code_amd.compat as nsc
    idx, _, _, _, hh = nsc.neighbors(p, np.inf, K)
    F2 = nsc.hydro_update(idx, p, s["mass"], hh, s["f_un"], s["particle_type"], s["T"], s["mu_array"], s["gamma_array"], v)[5]
    np.testing.assert_allclose(got["f_un_neighbor"], F2, rtol=1e-14, atol=0)
    monkeypatch.setenv("SPHX_SPECIES_LDS", "0")
    sim_g = Simulation(s, n_neigh=K, with_species=True, agb=table)
    sim_g.step(1)
    got_g = sim_g.download_species()
    assert np.array_equal(F2, got_g["f_un_neighbor"])
    np.testing.assert_allclose(got["metallicity"][fin], got_g["metallicity"][fin], rtol=1e-13)


@pytest.mark.gpu
def test_species_pass_inside_pass_1_equals_the_two_kernels(monkeypatch):
    """In the step loop under hydro_update's sums the species pass (nsc:624-627, + metallicity drv:663 and the AGB yields of
    config_helper.py:138-189) runs inside pass 1's kernel - its first sweep repeats pass 1's staging and kernel values
    (sphx_blob.hip blob_density_species_kernel).  Same expressions, same order: three steps of a two-phase cloud give the
    same bits as the two kernels run one after the other (SPHX_SPECIES_FUSED=0)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "agb_reference.npz"))
    import sph_code_amd.ics as ics
    import sph_code_amd.agb as agb
    from sph_code_amd.sim import Simulation
    s = ics.two_phase(30000)
    table = agb.splines_from_arrays(g["tx"], g["ty"], g["coeffs"], g["mapto"], float(g["divisor"]))
    res = {}
    for name, env in (("fused", None), ("separate", "0")):
        if env is not None:
            monkeypatch.setenv("SPHX_SPECIES_FUSED", env)
        sim = Simulation(s, n_neigh=40, with_species=True, agb=table)
        sim.step(3)
        res[name] = (sim.download(), sim.download_species())
        if env is not None:
            monkeypatch.delenv("SPHX_SPECIES_FUSED")
    for k in ("points", "velocities", "densities", "num_densities", "total_accel", "E_internal", "sizes"):
        assert np.array_equal(res["fused"][0][k], res["separate"][0][k], equal_nan=True), k
    for k in ("f_un_neighbor", "metallicity", "agb_dust"):
        assert np.array_equal(res["fused"][1][k], res["separate"][1][k], equal_nan=True), k
    assert np.isfinite(res["fused"][1]["f_un_neighbor"]).all() and (res["fused"][1]["f_un_neighbor"] > 0).any()
