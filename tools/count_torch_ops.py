"""Which lines of the decomposed step launch the small torch kernels (run on the GPU box):
   SPHX_FORCE_DIST=1 python3 tools/count_torch_ops.py  -> callers of zeros / zero_ / fill_ / full / clone / contiguous / to / cat
   per step of a one-rank rehearsal of bench.py's distributed path."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
counts = collections.Counter()
active = [False]
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        if active[0]:
            st = traceback.extract_stack(limit=4)
            for fr in reversed(st[:-1]):
                if "multigpu" in fr.filename:
                    counts[(name, fr.lineno, fr.line.strip()[:90])] += 1
                    break
        return orig(*a, **k)
    setattr(owner, name, f)
for n in ("zeros", "full", "zeros_like", "cat", "empty", "tensor", "stack", "clamp", "arange"):
    wrap(torch, n)
for n in ("zero_", "fill_", "clone", "contiguous", "to", "copy_", "any", "max", "sum", "abs", "tolist", "item", "__float__", "__getitem__"):
    wrap(torch.Tensor, n)
os.environ["SPHX_FORCE_DIST"] = "1"
sys.argv = ["bench.py", "--no-cpu", "--steps", "6", "--warmup", "3"]
import sph_code_amd.multigpu as mg
orig_step = mg.DistributedSim.step
nstep = [0]
def step(self, *a, **k):
    nstep[0] += 1
    active[0] = nstep[0] > 4
    r = orig_step(self, *a, **k)
    active[0] = False
    return r
mg.DistributedSim.step = step
import runpy
try:
    runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
except SystemExit:
    pass
n = max(nstep[0] - 4, 1)
print("steps counted:", n, file=sys.stderr)
for (name, line, src), c in counts.most_common(60):
    print("%5.1f/step  %-12s multigpu.py:%d  %s" % (c / n, name, line, src), file=sys.stderr)
