"""The CPU oracle (oracle/sph_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py).  Tolerances: SURVEY.md 8c."""
import numpy as np
import pytest

from conftest import hydro_args
from oracle import sph_oracle as orc

RTOL_POS = 1e-13      # sums of <= K non-negative terms
RTOL_SIGNED = 1e-10   # signed sums, max-norm


def test_constants_match_fixture(golden):
    assert golden["const_k"] == orc.K_B
    assert golden["const_amu"] == orc.AMU
    assert golden["const_m_h"] == orc.M_H
    assert golden["const_m_0"] == orc.M_0
    assert golden["const_dt_0"] == orc.DT_0
    np.testing.assert_array_equal(golden["mu_specie"], orc.MU_SPECIE)
    np.testing.assert_array_equal(golden["gamma_specie"], orc.GAMMA_SPECIE)
    np.testing.assert_allclose(orc.grain_mass(), golden["grain_mass"], rtol=1e-14)
    np.testing.assert_allclose(orc.sigma_effective(), golden["sigma_effective"], rtol=1e-14)


def test_neighbors_reference_mode(golden):
    """eps=0.1 (nsc:544): same SciPy call -> identical output."""
    K = int(golden["K"])
    idx, _, d, nt, h = orc.neighbors(golden["points"], float(golden["dist_bound"]), K, eps=0.1)
    np.testing.assert_array_equal(idx, golden["nb_idx"])
    np.testing.assert_array_equal(d, golden["nb_dist"])
    np.testing.assert_array_equal(nt, golden["nb_nontriv"])
    np.testing.assert_array_equal(h, golden["nb_h"])


def test_neighbors_bounded(golden):
    """distance_upper_bound active: missing neighbours idx == N, dist inf -> 0 (nsc:545-548)."""
    K = int(golden["K"])
    idx, _, d, nt, h = orc.neighbors(golden["points"], float(golden["bounded_r"]), K, eps=0.1)
    np.testing.assert_array_equal(idx, golden["bounded_idx"])
    np.testing.assert_array_equal(d, golden["bounded_dist"])
    np.testing.assert_array_equal(nt, golden["bounded_nontriv"])
    np.testing.assert_array_equal(h, golden["bounded_h"])
    assert (idx == len(idx)).any()


def test_exact_knn_bruteforce_vs_tree(golden):
    """Exact mode (eps=0): brute force == SciPy tree, index sets equal, distances rtol 1e-15."""
    K = int(golden["K"])
    idx, _, d, nt, h = orc.neighbors_bruteforce(golden["points"], np.inf, K)
    np.testing.assert_allclose(d, golden["exact_dist"], rtol=2e-15, atol=0)
    same = np.sort(idx, axis=1) == np.sort(golden["exact_idx"].astype(np.int64), axis=1)
    assert same.all()
    assert (idx[:, 0] == np.arange(len(idx))).all()


def test_approx_guarantee(golden):
    """h_exact <= h_ref <= 1.1 h_exact."""
    h_exact = golden["exact_dist"].max(axis=1)
    assert (golden["nb_h"] >= h_exact).all()
    assert (golden["nb_h"] <= 1.1 * h_exact * (1 + 1e-15)).all()


def _scales(g):
    """sum_k |term_k| bounds for the signed sums, from the oracle's own pair terms."""
    out, inter = orc.hydro_update(*hydro_args(g), return_intermediates=True)
    return out, inter


def test_hydro_update_captured_locals(golden):
    """Unmodified reference arithmetic nsc:556-649 (locals captured at the IndexError of nsc:651)."""
    g = golden
    (ha, va, vh, rho, nden, F, rho_d), inter = _scales(g)
    np.testing.assert_allclose(rho, g["cap_density_calc"], rtol=RTOL_POS)
    np.testing.assert_allclose(rho_d, g["cap_dust_density_calc"], rtol=RTOL_POS, atol=0)
    np.testing.assert_allclose(nden, g["cap_num_density_calc"], rtol=RTOL_POS)
    np.testing.assert_allclose(F, g["cap_f_un_neighbor"], rtol=RTOL_POS)
    G = inter["pressure_grad_symmetrized"]
    ref = g["cap_pressure_grad_symmetrized"]
    assert np.max(np.abs(G - ref)) <= RTOL_SIGNED * np.max(np.abs(ref))
    fin = np.isfinite(g["cap_hydro_accel"])
    assert np.max(np.abs(ha - g["cap_hydro_accel"])[fin]) <= RTOL_SIGNED * np.max(np.abs(g["cap_hydro_accel"][fin]))
    if "cap_w_ab" in g:
        np.testing.assert_allclose(inter["W6_kernel"], g["cap_W6_kernel"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(inter["distances"], g["cap_distances"], rtol=1e-15)
        np.testing.assert_array_equal(inter["neigh_sizes"], g["cap_neigh_sizes"])
        np.testing.assert_allclose(inter["w_ab"].T, g["cap_w_ab"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(inter["rho_avg_ab"].T, g["cap_rho_avg_ab"], rtol=RTOL_POS)
        np.testing.assert_allclose(inter["c_sound_ab"].T, g["cap_c_sound_ab"], rtol=1e-14)


def test_hydro_update_axis_repaired(golden):
    """All seven outputs of nsc:671 with the one-token repair of nsc:649."""
    g = golden
    ha, va, vh, rho, nden, F, rho_d = orc.hydro_update(*hydro_args(g))
    np.testing.assert_allclose(rho, g["hu_density_calc"], rtol=RTOL_POS)
    np.testing.assert_allclose(rho_d, g["hu_dust_density_calc"], rtol=RTOL_POS, atol=0)
    np.testing.assert_allclose(nden, g["hu_num_density_calc"], rtol=RTOL_POS)
    np.testing.assert_allclose(F, g["hu_f_un_neighbor"], rtol=RTOL_POS)
    for x, nm in ((ha, "hu_hydro_accel"), (va, "hu_visc_accel"), (vh, "hu_visc_heat")):
        ref = g[nm]
        fin = np.isfinite(ref)
        assert (np.isfinite(x) == fin).all(), nm
        assert np.max(np.abs(x - ref)[fin]) <= RTOL_SIGNED * np.max(np.abs(ref[fin])), nm


def test_loop_forms(golden):
    """nsc:673-816 with the injected global d."""
    g = golden
    d = float(g["loop_d"])
    nb = g["nb_idx"].astype(np.int64)
    P, m, pt, h = g["points"], g["mass"], g["particle_type"], g["nb_h"]
    rho = orc.density(P, m, pt, nb, d)
    np.testing.assert_allclose(rho, g["loop_density"], rtol=1e-12)
    np.testing.assert_allclose(orc.dust_density(P, m, nb, pt, h), g["loop_dust_density"], rtol=1e-12)
    np.testing.assert_allclose(orc.num_dens(m, P, g["mu_array"], nb, d), g["loop_num_dens"], rtol=1e-12)
    dp = orc.del_pressure(P, m, pt, nb, g["E_internal"], g["gamma_array"], d)
    assert np.max(np.abs(dp - g["loop_del_pressure"])) <= RTOL_SIGNED * np.max(np.abs(g["loop_del_pressure"]))
    acc, heat = orc.artificial_viscosity(nb, P, pt, h, m, g["loop_density"], g["velocities"], g["T"],
                                         g["gamma_array"], g["mu_array"], d)
    for x, ref in ((acc, g["loop_av_accel"]), (heat, g["loop_av_heat"])):
        fin = np.isfinite(ref)
        assert (np.isfinite(x) == fin).all()
        assert np.max(np.abs(x - ref)[fin]) <= RTOL_SIGNED * np.max(np.abs(ref[fin]))
    ct = orc.crossing_time(nb, g["velocities"], h, pt)
    assert ct == pytest.approx(float(g["loop_crossing_time"]), rel=1e-14)
    onto, react = orc.net_impulse(P, m, h, g["velocities"], pt, nb, g["f_un"])
    for x, ref in ((onto, g["loop_drag_onto"]), (react, g["loop_drag_reaction"])):
        sc = max(np.max(np.abs(ref)), 1e-300)
        assert np.max(np.abs(x - ref)) <= 1e-11 * sc


def test_sign_convention(golden):
    """hydro_update's accel is +grad P / rho; the loop form / rho has the physical sign."""
    g = golden
    gas = g["particle_type"] == 0
    ha = g["hu_hydro_accel"][gas]
    lp = (g["loop_del_pressure"] / g["loop_density"][:, None])[gas]
    ok = np.isfinite(ha).all(axis=1) & np.isfinite(lp).all(axis=1)
    cosang = np.sum(ha[ok] * lp[ok], axis=1)
    assert np.mean(cosang < 0) > 0.8


# ---- integrator + dt rule against the reference DRIVER's own statements (drv:222-238, 460-491) ----
def _driver_fixture():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "driver_integrator.npz")
    return np.load(path)


def test_driver_constants():
    z = _driver_fixture()
    assert float(z["const_dt_0"]) == orc.DT_0 and float(z["const_MAX_AGE"]) == orc.MAX_AGE
    assert float(z["const_AU"]) == orc.AU and float(z["const_m_h"]) == orc.M_H and float(z["const_k"]) == orc.K_B


def test_timestep_rule_vs_driver_block():
    """drv:222-231 executed on seeded (age, ct) pairs: first step, both clamps, ct > MAX_AGE (also DBL_MAX, inf)."""
    z = _driver_fixture()
    for age, ct, dt in z["dt_cases"]:
        assert orc.timestep(ct, age == 0) == dt, (age, ct, dt)


def test_clamp_vs_driver_block():
    """drv:233-238 on positions beyond +-1e11 AU, +-inf, NaN and non-finite velocities: bit-equal."""
    z = _driver_fixture()
    p, v = orc.clamp_state(z["clamp_points_in"], z["clamp_velocities_in"])
    assert np.array_equal(p, z["clamp_points_out"]) and np.array_equal(v, z["clamp_velocities_out"])


@pytest.mark.parametrize("case", ["gas", "dusty", "dusty_long", "first_shape", "maxage", "gas_nograv"])
def test_leapfrog_vs_driver_block(case):
    """drv:460-491 executed on seeded arrays (limiter tripped on a fifth of the particles, rho = 0,
    NaN heat/E, mismatched old_accel shape): assembly, limiter, leapfrog, E and T bit-equal."""
    z = _driver_fixture()
    g = lambda k: z["lf_%s_in_%s" % (case, k)]
    o = lambda k: z["lf_%s_out_%s" % (case, k)]
    dusty = case in ("dusty", "dusty_long", "maxage")
    drag = (g("drag_on_gas"), g("drag_reaction"))
    pa, visc = orc.assemble_loop(g("delp"), g("densities"), g("dust_densities"), g("particle_type"), g("av_accel"),
                                 drag)
    assert np.array_equal(pa, o("pressure_accel"))
    p, v, tot, E, T = orc.leapfrog(g("points"), g("velocities"), g("total_accel"), g("E_internal"), g("mass"),
                                   g("mu_array"), g("gamma_array"), pa, visc, g("av_heat"), float(g("dt")),
                                   g("grav_accel"))
    assert np.array_equal(tot, o("total_accel"))
    assert np.array_equal(p, o("points")) and np.array_equal(v, o("velocities"))
    assert np.array_equal(E, o("E_internal")) and np.array_equal(T, o("T"), equal_nan=True)
    if not dusty:        # without drag the pre-limiter viscous acceleration is av[0] alone (drv:473 adds zeros)
        pa2, visc2 = orc.assemble_loop(g("delp"), g("densities"), g("dust_densities"), g("particle_type"),
                                       g("av_accel"), None)
        t2 = orc.leapfrog(g("points"), g("velocities"), g("total_accel"), g("E_internal"), g("mass"),
                          g("mu_array"), g("gamma_array"), pa2, visc2, g("av_heat"), float(g("dt")), g("grav_accel"))[2]
        assert np.array_equal(t2, o("total_accel"))
