"""CPU oracle for the AGB dust-yield lookup - TEST INFRASTRUCTURE ONLY (see oracle/sph_oracle.py).

Restates sph/config_helper.py:138-211.  The fit itself lives in a third-party dependency: SciPy's
RectBivariateSpline (FITPACK regrid/surfit, SciPy 1.15 here; the reference pins no version) with kx = ky =
1 and s = 0.9, which on these tables returns no interior knots - each of the 11 yields becomes ONE bilinear
patch over [Z_min, Z_max] x [M_min, M_max] (4 coefficients).  Evaluation is restated in NumPy from
FITPACK's fpbisp/fpbspl (arguments clamped to the knot range, de Boor weights of degree 1), so it does not
need SciPy.  Pinned against outputs of the reference's own two functions run in the build container
(tests/golden/agb_reference.npz, made by tests/golden/make_golden_agb.py).
"""
import os

import numpy as np

SOLAR_MASS = 1.989e30                                   # nsc:24
AGB_MASSES = np.array([1, 1.1, 1.2, 1.25, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8, 1.9, 2.0, 2.1, 2.2, 2.3, 2.4, 2.5, 3, 3.5,
                       4, 4.01, 4.5, 5.0, 5.5, 6.0, 6.5, 7])                        # config_helper.py:148
METALLICITY = np.array([0.001, 0.002, 0.004, 0.008, 0.015, 0.02, 0.03, 0.04])      # config_helper.py:149
MAPTO = np.array([6, 6, 11, 12, 7, 10, 7, 13, 8, 10, 13])                          # config_helper.py:168
AGB_DIVISOR = 3                                                                     # config_helper.py:169


def load_tables(path, sort_files=True):
    """config_helper.py:152-166: one file per metallicity -> (species, metallicity, mass); values <= 0
    become 1e-30.  The reference takes os.listdir order (arbitrary); sorted is what its labels assume."""
    files = [f for f in os.listdir(path) if f != ".DS_Store"]
    if sort_files:
        files = sorted(files)
    by_met = [np.genfromtxt(os.path.join(path, f)).T[1:] for f in files]
    seg = np.swapaxes(by_met, 0, 1).copy()
    seg[seg <= 0.] = 1e-30
    return seg


def fit(tables, s=0.9):
    """config_helper.py:172-176.  Returns (tx, ty, coeffs) lists, one entry per yield."""
    from scipy.interpolate import RectBivariateSpline
    tx, ty, cf = [], [], []
    for item in range(tables.shape[0]):
        sp = RectBivariateSpline(METALLICITY, AGB_MASSES, tables[item], kx=1, ky=1, s=s)
        kx_, ky_ = sp.get_knots()
        tx.append(np.asarray(kx_)); ty.append(np.asarray(ky_)); cf.append(np.asarray(sp.get_coeffs()))
    return tx, ty, cf


def _weights(t, x):
    """FITPACK fpbisp/fpbspl for degree 1: clamp, locate the knot interval, two de Boor weights."""
    n = len(t)
    x = np.minimum(np.maximum(np.asarray(x, dtype=np.float64), t[1]), t[n - 2])
    l = np.clip(np.searchsorted(t, x, side="right"), 2, n - 2)      # t[l-1] <= x < t[l] (1-based l)
    f = 1.0 / (t[l] - t[l - 1])
    return l - 2, f * (t[l] - x), f * (x - t[l - 1])


def evaluate(tx, ty, c, x, y):
    """Tensor-product degree-1 spline at the point pairs (x_i, y_i)."""
    ny = len(ty) - 2
    c = np.asarray(c).reshape(len(tx) - 2, ny)
    ix, wx0, wx1 = _weights(tx, x)
    iy, wy0, wy1 = _weights(ty, y)
    sp = c[ix, iy] * wx0 * wy0
    sp = sp + c[ix, iy + 1] * wx0 * wy1
    sp = sp + c[ix + 1, iy] * wx1 * wy0
    sp = sp + c[ix + 1, iy + 1] * wx1 * wy1
    return sp


def calculate_interpolation(masses, metallicities, splines, mu_specie, composition, mapto=MAPTO,
                            divisor=AGB_DIVISOR):
    """config_helper.py:180-211.  splines = (tx, ty, coeffs).  Returns (dust (n,S), gas (n,S))."""
    tx, ty, cf = splines
    masses = np.asarray(masses, dtype=np.float64)
    n, S = len(masses), len(mu_specie)
    dust = np.zeros((n, S))
    mass_wd = (0.55 + (masses / SOLAR_MASS - 1.) * 0.45 / (7. - 1.)) * SOLAR_MASS
    for o in range(len(cf)):                      # fancy assignment with repeated targets: the last wins
        dust[:, mapto[o]] = evaluate(tx[o], ty[o], cf[o], metallicities, masses)
    dust /= divisor
    dust[dust < 0.] = 0.
    gas_mass = masses - np.sum(dust, axis=1) - mass_wd
    num = (np.asarray(composition, dtype=np.float64) / mu_specie).T.copy()
    ion, h2, h = num[3] * 0.1, num[0] * 0.1, num[2] * 0.1
    num[0] -= h2
    num[1] += h / 4. + h2 / 2. + ion / 4.
    num[2] -= h
    num[3] -= ion
    num[5] -= ion
    gm = (num.T * mu_specie).T / np.sum(num.T * mu_specie, axis=1)
    return dust, (gm * gas_mass).T
