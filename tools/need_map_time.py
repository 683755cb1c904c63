"""Time sphx_dev_need_map on a synthetic owned set shaped like the evolved uniform cube (run on the GPU box)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sph_code_amd.multigpu import LibBackend
be = LibBackend(0, k=40)
dev = be.device
rs = np.random.RandomState(1)
n, G = 500000, 96
L = 1.87e17
cs = 2.9e15
lo = [-0.5 * G * cs] * 3
pos = (rs.rand(n, 3) - 0.5) * L
pos[:, 0] = -np.abs(pos[:, 0])                      # one half of the cube
nrim = 5000
pos[:nrim] *= 1.0 + rs.rand(nrim, 1) * 0.6          # a rim outside
h = np.full(n, 4.2e15)
h[:nrim] = 10.0 ** rs.uniform(np.log10(1.5e16), np.log10(1.9e17), nrim)
w = 1.3 * h
P = torch.as_tensor(pos, device=dev); W = torch.as_tensor(w, device=dev)
for wide in (True, False):
    Wt = W if wide else torch.clamp(W, max=1.3 * 4.2e15)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = be.need_map(P, Wt, lo, cs, G)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print("wide claims" if wide else "narrow only", "need_map %.3f ms, marked %d of %d cells" % ((t1 - t0) * 1e3, int(m.sum()), G ** 3))
