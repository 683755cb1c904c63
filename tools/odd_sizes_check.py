import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import sph_code_amd.ics as ics
from sph_code_amd.sim import Simulation
n, K, wl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
s0 = ics.WORKLOADS[wl](n)
kw = dict(forms="loop", d=ics.loop_d(s0, K)) if wl == "uniform_cube" else {}
sim = Simulation(s0, n_neigh=K, **kw)
sim.step(6)
d = sim.download()
import hashlib
h = hashlib.sha256()
for k in ("points", "velocities", "sizes", "densities", "E_internal"):
    h.update(np.ascontiguousarray(d[k]).tobytes())
print(h.hexdigest(), sim.stats()["fallback_queries"])
'''
open("/tmp/sphx_odd_child.py", "w").write(code)
for n, K, wl in [(100, 7, "polytrope"), (1000, 40, "polytrope"), (1537, 40, "uniform_cube"), (129, 40, "polytrope"), (5000, 64, "polytrope"), (777, 1, "uniform_cube")]:
    out = {}
    for name, env in (("grouped", {}), ("general", {"SPHX_KNN_GROUP": "0"}), ("nomfma_tie0", {"SPHX_TIE_FIX": "0"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "/tmp/sphx_odd_child.py", str(n), str(K), wl], env=e, capture_output=True, text=True)
        out[name] = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-200:]
    ok = out["grouped"].split()[0] == out["general"].split()[0] == out["nomfma_tie0"].split()[0]
    print(n, K, wl, "OK" if ok else "MISMATCH", out)
