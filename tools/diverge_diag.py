"""Per-step cost and state spread while the reference's hydro_update scheme diverges on a dense cloud (run on the GPU box):
   python3 tools/diverge_diag.py [workload] [n] [steps]
Prints, per step: ms by section, candidates / retries / fallback queries, cells and cell edge, and percentiles of
|x - centre| and h - what the search has to cope with once escapers have been clamped into the boundary cells."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the library: see INTEGRATION.md)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sph_code_amd.ics as ics  # noqa: E402
from sph_code_amd.sim import Simulation  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "uniform_cube"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 14
s0 = ics.WORKLOADS[wl](n)
sim = Simulation(s0, n_neigh=40)
sim.ctx.set_timing_detail(True)
q = [50, 90, 99, 99.9, 100]
for it in range(steps):
    sim.reset_stats()
    sim.step(1)
    st = sim.stats()
    line = "step %2d total %8.2f grid %6.2f search %8.2f sums %5.2f | cand/q %7.1f retries %7d fallback %7d short %5d far %6d levels %2d cells %8d cs %.3g" % (
        it, st["ms_total"], st["ms_grid"], st["ms_search"], st["ms_prep"] + st["ms_density"] + st["ms_pi"] + st["ms_visc"],
        st["candidates"] / float(n), st["retries"], st["fallback_queries"], st["short_rows"], st["far_queries"], st["outlier_levels"], st["cells"], st["cell_size"])
    if it % 2 == 1 or it == steps - 1:
        d = sim.download()
        p = d["points"]
        c = np.median(p, axis=0)
        r = np.abs(p - c).max(axis=1)
        h = d["sizes"]
        fin = np.isfinite(h)
        big = np.abs(p).max()
        at = (np.abs(p) == big)
        uq, cnt = np.unique(p, axis=0, return_counts=True)
        line += "\n        max|x| %.6g: particles with 1/2/3 coordinates there %d %d %d; distinct positions %d, largest pile-ups %s" % (
            big, (at.sum(1) == 1).sum(), (at.sum(1) == 2).sum(), (at.sum(1) == 3).sum(), len(uq), np.sort(cnt)[-10:])
        line += "\n        |x-c|_inf pct%s = %s\n        h pct = %s  (finite %d)  max|v| %.3g" % (
            q, np.array2string(np.percentile(r, q), precision=3), np.array2string(np.percentile(h[fin], q), precision=3),
            fin.sum(), np.nanmax(np.abs(d["velocities"])))
        # in-box / out-of-box split against a box reconstructed from the cell count (cube-shaped grids only: a rough guide)
        half = 0.5 * st["cells"] ** (1. / 3.) * st["cell_size"]
        inb = r <= half
        rc = 1.08 * h / st["cell_size"]
        for nm, sel in (("in-box", inb), ("out-of-box", ~inb)):
            if sel.sum():
                line += "\n        %-10s %7d particles; search radius in cells pct%s = %s; > 8: %d, > 30: %d, > 100: %d" % (
                    nm, sel.sum(), q, np.array2string(np.percentile(rc[sel], q), precision=1), (rc[sel] > 8).sum(),
                    (rc[sel] > 30).sum(), (rc[sel] > 100).sum())
        # how far each particle moved this step, in units of its own h (what the hinted radius has to absorb)
        if "prev" in globals() and prev is not None:
            mv = np.linalg.norm(p - prev, axis=1) / np.maximum(h, 1e-300)
            line += "\n        displacement / h pct = %s" % np.array2string(np.percentile(mv[np.isfinite(mv)], q), precision=2)
    prev = sim.download()["points"] if (it % 2 == 0 and it + 1 < steps) else None
    print(line, flush=True)
