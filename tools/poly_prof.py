import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sph_code_amd.ics as ics
from sph_code_amd.sim import Simulation
s0 = ics.WORKLOADS["polytrope"](1000000)
sim = Simulation(s0, n_neigh=40)
for it in range(6):
    sim.reset_stats()
    sim.step(1)
    st = sim.stats()
    print("step", it, "search %.3f" % st["ms_search"], "fallback", st["fallback_queries"], flush=True)
