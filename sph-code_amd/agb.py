"""AGB dust-yield lookup: drop-in for the two functions of sph/config_helper.py that the reference's
post-processing step calls per batch of AGB stars (config_helper.py:138-211).

    splines, mapto, AGB_divisor = interpolate_amounts(absolute_path_to_nsc)
    dust, gas = calculate_interpolation(AGB_masses, AGB_metallicities, splines, mapto, AGB_divisor,
                                        mu_specie, AGB_composition)

`interpolate_amounts` is host set-up, as in the reference: it reads the eight Z_*.txt tables and fits the 11
yields with SciPy's RectBivariateSpline(kx=1, ky=1, s=0.9) - the fit is a third-party algorithm (FITPACK)
on an 8 x 27 table, run once.  `calculate_interpolation` - one evaluation of every spline per star, a
Python loop in the reference - runs on the GPU (csrc/sphx_agb.hip through sphx_agb_yields).

Reference quirks kept: masses are passed in kg while the table's mass axis is in solar masses,
so FITPACK clamps every lookup to the 7 M_sun edge; `mapto` has repeated targets and the last spline wins;
s = 0.9 collapses each yield to a single bilinear patch (pass s=0 for a true piecewise-bilinear table).
The reference lists the table directory in arbitrary os.listdir order; sorted order (the default here) is
what its metallicity labels assume.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

solar_mass = 1.989e30                                                        # nsc:24
AGB_masses = np.array([1, 1.1, 1.2, 1.25, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8, 1.9, 2.0, 2.1, 2.2, 2.3, 2.4, 2.5, 3, 3.5,
                       4, 4.01, 4.5, 5.0, 5.5, 6.0, 6.5, 7])                 # config_helper.py:148
metallicity = np.array([0.001, 0.002, 0.004, 0.008, 0.015, 0.02, 0.03, 0.04])   # config_helper.py:149


class Spline:
    """Knots and coefficients of one fitted yield (degree 1 in both directions)."""
    def __init__(self, tx, ty, coeffs):
        self.tx = np.ascontiguousarray(tx, dtype=np.float64)
        self.ty = np.ascontiguousarray(ty, dtype=np.float64)
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.float64).ravel()
        if self.coeffs.size != (self.tx.size - 2) * (self.ty.size - 2):
            raise ValueError("coefficient count does not match the knots of a degree-1 spline")

    def get_knots(self):
        return self.tx, self.ty

    def get_coeffs(self):
        return self.coeffs


def load_tables(absolute_path_to_AGB, sort_files=True):
    """config_helper.py:152-166 -> (species, metallicity, mass) with values <= 0 set to 1e-30."""
    names = os.listdir(absolute_path_to_AGB)
    if sort_files:
        names = sorted(names)
    cube = None                                           # (metallicity, species, mass) while reading
    for name in names:
        if name == ".DS_Store":
            continue
        yields = np.genfromtxt(os.path.join(absolute_path_to_AGB, name))[:, 1:].T        # column 0 is the stellar mass
        cube = yields[None] if cube is None else np.concatenate([cube, yields[None]], axis=0)
    table = np.transpose(cube, (1, 0, 2)).copy()
    return np.where(table <= 0., 1e-30, table)


def fit_tables(species_segregated, s=0.9):
    from scipy.interpolate import RectBivariateSpline
    out = []
    for item in range(species_segregated.shape[0]):
        sp = RectBivariateSpline(metallicity, AGB_masses, species_segregated[item], kx=1, ky=1, s=s)
        tx, ty = sp.get_knots()
        out.append(Spline(tx, ty, sp.get_coeffs()))
    return out


def splines_from_arrays(tx, ty, coeffs, mapto, divisor=3):
    """(splines, mapto, divisor) from stored knots / coefficients (one row per spline), e.g. the data a previous
    interpolate_amounts produced - the form tests/golden/agb_reference.npz keeps them in."""
    return [Spline(a, b, c_) for a, b, c_ in zip(tx, ty, coeffs)], np.asarray(mapto), divisor


def interpolate_amounts(absolute_path_to_nsc, s=0.9, sort_files=True):
    """config_helper.py:138-178: (splines, mapto, AGB_divisor)."""
    tables = load_tables(os.path.join(absolute_path_to_nsc, "..", "agb_interp"), sort_files=sort_files)
    mapto = np.array([6, 6, 11, 12, 7, 10, 7, 13, 8, 10, 13])                 # config_helper.py:168
    return fit_tables(tables, s=s), mapto, 3


def calculate_interpolation(AGB_masses, AGB_metallicities, splines, mapto, AGB_divisor, mu_specie,
                            AGB_composition=None, ctx=None):
    """config_helper.py:180-211 on the GPU: (dust_mass_created (n,S), gas mass by species (n,S)); the
    second is None when AGB_composition is None."""
    c = ctx if ctx is not None else _lib.default_context()
    m = np.ascontiguousarray(AGB_masses, dtype=np.float64).ravel()
    z = np.ascontiguousarray(AGB_metallicities, dtype=np.float64).ravel()
    if m.shape != z.shape:
        raise ValueError("AGB_masses and AGB_metallicities differ in length")
    mu = np.ascontiguousarray(mu_specie, dtype=np.float64).ravel()
    n, S = m.size, mu.size
    if n == 0:
        return np.zeros((0, S)), (None if AGB_composition is None else np.zeros((0, S)))
    ntx = np.array([sp.get_knots()[0].size for sp in splines], dtype=np.int32)
    nty = np.array([sp.get_knots()[1].size for sp in splines], dtype=np.int32)
    tx = np.ascontiguousarray(np.concatenate([np.asarray(sp.get_knots()[0], dtype=np.float64) for sp in splines]))
    ty = np.ascontiguousarray(np.concatenate([np.asarray(sp.get_knots()[1], dtype=np.float64) for sp in splines]))
    cf = np.ascontiguousarray(np.concatenate([np.asarray(sp.get_coeffs(), dtype=np.float64).ravel() for sp in splines]))
    mp = np.ascontiguousarray(mapto, dtype=np.int32)
    if mp.size != len(splines):
        raise ValueError("mapto must name one species per spline")
    dust = np.empty((n, S))
    comp = gas = None
    if AGB_composition is not None:
        comp = np.ascontiguousarray(AGB_composition, dtype=np.float64)
        if comp.shape != (n, S):
            raise ValueError("AGB_composition must be (n, %d)" % S)
        gas = np.empty((n, S))
    dp = lambda a: None if a is None else a.ctypes.data_as(_lib.c_double_p)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    c.check(c.lib.sphx_agb_yields(c.h, n, dp(m), dp(z), len(splines), ip(ntx), ip(nty), dp(tx), dp(ty), dp(cf),
                                  ip(mp), float(AGB_divisor), S, dp(mu), dp(comp), solar_mass, dp(dust), dp(gas)))
    return dust, gas
