"""CPU-side checks of the drop-in boundary: libsphx.so loads and exports every symbol
include/sphx.h declares; the product path fails loudly without a GPU (no CPU fallback);
the compat module mirrors the reference's signatures."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sphx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sphx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import sph_code_amd._lib as L
    lib = L.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(lib, nm), "libsphx.so does not export %s" % nm
    assert set(names) == set(L.SIGNATURES), set(names) ^ set(L.SIGNATURES)
    assert lib.sphx_version() >= 100


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sph_code_amd._lib as L
    with pytest.raises(RuntimeError):
        L.Context(0)
    import sph_code_amd.compat as nsc
    with pytest.raises(RuntimeError):
        nsc.neighbors(np.random.rand(10, 3), 1.0, 4)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "sph-code_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle bit for bit", ""), (dirpath, f)


def test_compat_signatures_match_reference():
    """Positional argument names and order of the reference callables."""
    import sph_code_amd.compat as nsc
    want = {
        "neighbors": ["points", "dist", "N_NEIGH"],
        "hydro_update": ["neighbor", "points", "mass", "sizes", "f_un", "particle_type", "T", "mu_array",
                         "gamma_array", "velocities"],
        "density": ["points", "mass", "particle_type", "neighbor"],
        "dust_density": ["points", "mass", "neighbor", "particle_type", "sizes"],
        "num_dens": ["mass", "points", "mu_array", "neighbor"],
        "del_pressure": ["points", "mass", "particle_type", "neighbor", "E_internal", "gamma_array"],
        "artificial_viscosity": ["neighbor", "points", "particle_type", "sizes", "mass", "densities",
                                 "velocities", "T", "gamma_array", "mu_array"],
        "crossing_time": ["neighbor", "velocities", "sizes", "particle_type"],
        "net_impulse": ["points", "mass", "sizes", "velocities", "particle_type", "neighbor", "f_un"],
    }
    for name, args in want.items():
        params = list(inspect.signature(getattr(nsc, name)).parameters)
        assert params[:len(args)] == args, (name, params)
    from oracle import sph_oracle as orc
    assert nsc.k == orc.K_B and nsc.amu == orc.AMU and nsc.m_h == orc.M_H and nsc.m_0 == orc.M_0
    assert nsc.dt_0 == orc.DT_0
    np.testing.assert_array_equal(nsc.mu_specie, orc.MU_SPECIE)
    np.testing.assert_array_equal(nsc.gamma, orc.GAMMA_SPECIE)
    np.testing.assert_allclose(nsc.grain_mass(), orc.grain_mass(), rtol=1e-15)
    np.testing.assert_allclose(nsc.sigma_effective(), orc.sigma_effective(), rtol=1e-15)


def test_ics_shapes_and_seeding():
    import sph_code_amd.ics as ics
    for name, fn in ics.WORKLOADS.items():
        a, b = fn(2000), fn(2000)
        assert a["points"].shape == (2000, 3) and a["f_un"].shape == (2000, 15)
        assert np.array_equal(a["points"], b["points"]), name
        assert np.isfinite(a["E_internal"]).all() and (a["T"] > 0).all()
    s = ics.sedov_sphere(20000)
    ek = 0.5 * np.sum(s["mass"][:, None] * s["velocities"] ** 2)
    assert 0.7e44 < ek < 0.8e44


def test_oracle_step_runs():
    """The CPU baseline leg of bench.py: one oracle step on a small sphere is finite."""
    from oracle import sph_oracle as orc
    import sph_code_amd.ics as ics
    s = ics.uniform_sphere(3000)
    r = orc.step(s, n_neigh=40, eps=0.0, first=True)
    assert np.isfinite(r["points"]).all() and np.isfinite(r["velocities"]).all()
    assert r["dt"] == orc.DT_0 / 10
    r2 = orc.step(r, n_neigh=40, eps=0.0, first=False)
    assert orc.DT_0 / 5 <= r2["dt"] <= 2 * orc.DT_0


def test_pinned_pool_falls_back_to_ordinary_memory_without_a_gpu():
    """compat hands large results back in page-locked buffers (sphx_host_alloc); where pages cannot be
    locked - no HIP device here - the pool must quietly return ordinary arrays of the right shape."""
    from sph_code_amd import _lib
    a = _lib.pinned.empty((3 << 20,), np.float64)           # above the pool's size threshold
    assert a.shape == (3 << 20,) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
    a[:4] = 1.0
    b = _lib.pinned.empty((10, 3), np.float64)              # small: never pooled
    assert b.shape == (10, 3)
    del a


def test_product_library_carries_no_experiment_switches():
    """Timing experiments and diagnostics that add launches or change results (SPHX_*_EXP*, SPHX_KNN_ABL, *_PROF,
    SPHX_KG_DEBUG) are compiled in only with -DSPHX_EXPERIMENTS; the shipped libsphx.so must not even hold their names,
    and says so in sphx_build_info."""
    import sph_code_amd._lib as L
    lib = L.load_library()
    if os.environ.get("SPHX_LIB") or "SPHX_EXPERIMENTS" in os.environ.get("SPHX_EXTRA_FLAGS", ""):
        pytest.skip("an experiment build was asked for")
    info = lib.sphx_build_info().decode()
    assert "experiments=0" in info and "knn_prof" not in info, info
    blob = open(L.LIB_PATH, "rb").read()
    for name in (b"SPHX_KG_EXP_NOAMB", b"SPHX_EXP_NO_AGB", b"SPHX_KNN_ABL", b"SPHX_BLOB_EXP", b"SPHX_PASS_EXP",
                 b"SPHX_KG_PROF", b"SPHX_KNN_PROF", b"SPHX_KG_DEBUG"):
        assert name not in blob, name
    assert re.search(rb"SPHX_[A-Z_]*EXP", blob) is None
    # the tunables that remain are read in ONE place, sphx_create
    for f in os.listdir(os.path.join(ROOT, "sph-code_amd", "csrc")):
        if f.endswith((".hip", ".h")) and f != "sphx_api.hip":
            src = open(os.path.join(ROOT, "sph-code_amd", "csrc", f)).read()
            src = re.sub(r"#ifdef SPHX_EXPERIMENTS.*?#e(ndif|lse)", "", src, flags=re.S)
            assert "getenv" not in src, f
