#!/usr/bin/env python3
"""Golden vectors for the AGB dust-yield lookup (sph/config_helper.py:138-211) from the reference.

CONTAINER-ONLY TOOL (no-op when /root/reference is absent).  config_helper.py is a script: importing
it would run a whole post-processing job on files that do not exist here.  Only the two function
definitions `interpolate_amounts` (config_helper.py:138-178) and `calculate_interpolation`
(config_helper.py:180-211) are taken - located with `ast` in the in-memory Python-2 -> 3 transform of
the file (tab expansion, lib2to3 fix_print / fix_unicode) - and executed in a namespace holding numpy,
os, copy, SciPy's RectBivariateSpline and the constants they use.  One change is applied to the
transformed text: `os.listdir(...)` is wrapped in `sorted(...)`; the reference takes the directory order,
which is arbitrary (sorted order is what the metallicity labels assume).
Nothing of the reference's source is stored: the fixture holds the AGB tables (data), the fitted
knots/coefficients, seeded inputs and the two functions' outputs.
"""
import ast
import copy
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/sph/config_helper.py"
NSC_DIR = "/root/reference/sph"
HERE = os.path.dirname(os.path.abspath(__file__))
SOLAR = 1.989e30


def load_functions():
    from lib2to3 import refactor
    warnings.simplefilter("ignore")
    src = open(REF).read().expandtabs(8)
    if not src.endswith("\n"):
        src += "\n"
    tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print", "lib2to3.fixes.fix_unicode"])
    src3 = str(tool.refactor_string(src, "config_helper"))
    tree = ast.parse(src3)
    lines = src3.split("\n")
    want = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("interpolate_amounts", "calculate_interpolation"):
            want[node.name] = "\n".join(lines[node.lineno - 1:node.end_lineno])
    assert len(want) == 2, list(want)
    text = want["interpolate_amounts"]
    assert text.count("os.listdir(absolute_path_to_AGB)") == 1
    want["interpolate_amounts"] = text.replace("os.listdir(absolute_path_to_AGB)",
                                                "sorted(os.listdir(absolute_path_to_AGB))")
    from scipy.interpolate import RectBivariateSpline
    ns = dict(np=np, os=os, copy=copy, RectBivariateSpline=RectBivariateSpline, solar_mass=SOLAR)
    for name in want:
        exec(compile(want[name], "config_helper:" + name, "exec"), ns)
    return ns["interpolate_amounts"], ns["calculate_interpolation"]


def main():
    if not os.path.exists(REF):
        print("make_golden_agb: /root/reference absent - nothing to do")
        return 0
    sys.stdout, keep = open(os.devnull, "w"), sys.stdout          # the functions print progress
    try:
        interp, calc = load_functions()
        splines, mapto, divisor = interp(NSC_DIR)
        rs = np.random.RandomState(77)
        n = 64
        # the reference passes stellar masses in kg (config_helper.py:182 divides by solar_mass),
        # while the table's mass axis is in solar masses: every lookup is clamped to the 7 M_sun edge.
        # The second half of the sample uses solar-mass-valued "masses" so the interior of the table
        # is exercised through the same function.
        masses = np.concatenate([rs.uniform(1.0, 7.0, n // 2) * SOLAR, rs.uniform(0.8, 7.5, n // 2)])
        metal = rs.uniform(0.0005, 0.05, n)
        # nsc:41 (molecular weights of the 15 tracked species)
        mu_specie = np.array([2.0158, 4.0026, 1.0079, 1.0074, 4.0021, 4.0016, 0.0005, 140.69, 60.08, 12.0107,
                              28.0855, 55.834, 100.39, 131.93, 40.096])
        comp = rs.uniform(0.01, 1.0, (n, 15))
        comp /= comp.sum(axis=1)[:, None]
        dust, gas = calc(masses, metal, splines, mapto, divisor, mu_specie, comp)
    finally:
        sys.stdout = keep
    # the tables themselves (data): species x metallicity x mass, as the reference stacks them
    files = sorted(f for f in os.listdir(NSC_DIR + "/../agb_interp") if f != ".DS_Store")
    tables = np.swapaxes([np.genfromtxt(NSC_DIR + "/../agb_interp/" + f).T[1:] for f in files], 0, 1)
    tx = np.array([s.get_knots()[0] for s in splines])
    ty = np.array([s.get_knots()[1] for s in splines])
    cf = np.array([s.get_coeffs() for s in splines])
    grid_z = np.array([0.0005, 0.001, 0.003, 0.01, 0.02, 0.04, 0.05])
    grid_m = np.array([0.9, 1.0, 1.25, 2.05, 3.3, 4.005, 6.9, 7.0, 8.0])
    grid_vals = np.array([[[float(s(z, m)[0, 0]) for m in grid_m] for z in grid_z] for s in splines])
    np.savez_compressed(os.path.join(HERE, "agb_reference.npz"), tables=tables, files=np.array(files),
                        mapto=np.asarray(mapto), divisor=float(divisor), tx=tx, ty=ty, coeffs=cf,
                        masses=masses, metallicities=metal, mu_specie=mu_specie, composition=comp,
                        dust_mass_created=dust, gas_mass_created=gas, grid_z=grid_z, grid_m=grid_m,
                        grid_vals=grid_vals)
    print("wrote agb_reference.npz: %d splines, knots %s/%s, sample %d" % (len(splines), tx.shape, ty.shape, n))
    return 0


if __name__ == "__main__":
    sys.exit(main())
