import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sph_code_amd.ics as ics
from sph_code_amd.sim import Simulation
for wl, n in (("uniform_sphere", 3000), ("polytrope", 20000)):
    s = ics.WORKLOADS[wl](n)
    sim = Simulation(s, n_neigh=40)
    for it in range(3):
        sim.step(1)
        d = sim.download()
        print(wl, it, sim.failures(), "finite accel:", np.isfinite(d["total_accel"]).all(), "heat finite", np.isfinite(d["visc_heat"]).mean(),
              "dt", d["dt"], "E finite", np.isfinite(d["E_internal"]).all(), flush=True)
