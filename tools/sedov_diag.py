"""Sedov blast (BASELINE configs[2]) under the Courant-limited step, step by step (run on the GPU box):
   python3 tools/sedov_diag.py [forms] [steps] [n]
ms per step and per search, the search's counters, the cloud's extent and speeds: where the blast stops being a blast."""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sph_code_amd.ics as ics
from sph_code_amd.sim import Simulation
forms = sys.argv[1] if len(sys.argv) > 1 else "hydro_update"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
s0 = ics.sedov_sphere(n)
dt = ics.cfl_dt(s0, 40)
kw = dict(forms="loop", d=ics.loop_d(s0, 40)) if forms == "loop" else {}
sim = Simulation(s0, n_neigh=40, **kw)
R0 = np.abs(s0["points"]).max()
print("sedov, %s sums, n %d, fixed dt %.4g s, initial radius %.4g m" % (forms, n, dt, R0), flush=True)
for it in range(steps):
    sim.reset_stats()
    sim.step(1, fixed_dt=dt)
    st = sim.stats()
    line = "step %2d total %8.3f search %8.3f | cand/q %7.1f retries %6d fallback %6d far %5d levels %2d cells %8d" % (
        it, st["ms_total"], st["ms_search"], st["candidates"] / float(n), st["retries"], st["fallback_queries"], st["far_queries"],
        st["outlier_levels"], st["cells"])
    if it % 5 == 4 or st["ms_total"] > 3.0:
        d = sim.download()
        r = np.linalg.norm(d["points"], axis=1)
        v = np.linalg.norm(d["velocities"], axis=1)
        h = d["sizes"]
        line += "\n        |x| pct[50 99 99.99 100] %s  |v| pct %s  h pct %s  beyond 3 R0: %d, at the clamp: %d; rho max/median %.3g" % (
            np.array2string(np.percentile(r, [50, 99, 99.99, 100]), precision=3), np.array2string(np.percentile(v, [50, 99, 99.99, 100]), precision=3),
            np.array2string(np.percentile(h, [50, 99, 99.99, 100]), precision=3), (r > 3 * R0).sum(), (np.abs(d["points"]).max(axis=1) >= 1.49e22).sum(),
            np.nanmax(d["densities"]) / np.nanmedian(d["densities"]))
        f = sim.failures()
        line += "\n        failures " + str(f)
    print(line, flush=True)
