import os
import sys

import numpy as np
import pytest

try:        # torch brings its own copy of the HIP runtime: in a process that uses both, it must be loaded BEFORE
    import torch  # noqa: F401  libsphx.so pulls in the system one (else torch later finds "No HIP GPUs")
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

GOLDEN_CASES = ["cube_gas_n2048_k40", "sphere_dust_n2048_k40", "condensed_n1024_k40",
                "converging_n1024_k16", "small_n256_k16"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    g = load_golden(request.param)
    g["_name"] = request.param
    return g


def hydro_args(g, neighbor=None, sizes=None):
    """Positional arguments of hydro_update in the reference's order (nsc:556)."""
    nb = g["nb_idx"].astype(np.int64) if neighbor is None else neighbor
    h = g["nb_h"] if sizes is None else sizes
    return (nb, g["points"], g["mass"], h, g["f_un"], g["particle_type"], g["T"], g["mu_array"],
            g["gamma_array"], g["velocities"])


def assert_signed_sum_close(x, ref, scale, tol=1e-12, what=""):
    """|x - ref| <= tol * sum_k|term_k| componentwise - `scale` is that sum (or a
    bound of it)."""
    x = np.asarray(x); ref = np.asarray(ref)
    bad = ~(np.abs(x - ref) <= tol * scale)
    bad &= ~(np.isnan(x) & np.isnan(ref))
    bad &= ~((x == ref))
    assert not bad.any(), "%s: %d elements off, worst %g" % (
        what, bad.sum(), np.nanmax(np.abs(x - ref)[bad] / scale[bad]))
