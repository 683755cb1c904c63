"""ms per step over a long run, 50 steps at a time (run on the GPU box):  python3 tools/long_run.py [workload] [forms] [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sph_code_amd.ics as ics
from sph_code_amd.sim import Simulation
wl = sys.argv[1] if len(sys.argv) > 1 else "polytrope"
forms = sys.argv[2] if len(sys.argv) > 2 else "loop"
total = int(sys.argv[3]) if len(sys.argv) > 3 else 400
n = 1000000
s0 = ics.WORKLOADS[wl](n, light=True)
kw = dict(forms="loop", d=ics.loop_d(s0, 40)) if forms == "loop" else {}
sim = Simulation(s0, n_neigh=40, **kw)
if os.environ.get("LONG_RUN_DETAIL"):
    sim.ctx.set_timing_detail(True)
sim.step(3)
done = 3
while done < total:
    sim.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.step(50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done += 50
    st = sim.stats()
    d = sim.download()
    h = d["sizes"]
    print("steps %4d-%4d  %.3f ms/step  search %.3f  cells %8d  fallback %6d far %6d levels %2d  h mean %.3g max %.3g  max|v| %.3g" % (
        done - 50, done, dt / 50 * 1e3, st["ms_search"] / max(st["steps"], 1), st["cells"], st["fallback_queries"], st["far_queries"],
        st["outlier_levels"], h.mean(), h.max(), np.abs(d["velocities"]).max()), flush=True)
    if st["detail_steps"]:
        ds = float(st["detail_steps"])
        print("        per pass: " + "  ".join("%s %.3f" % (k_[3:], st[k_] / ds) for k_ in
              ("ms_grid", "ms_prep", "ms_density", "ms_pi", "ms_visc", "ms_integrate")), flush=True)
