"""Build libsphx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsphx.so")
SOURCES = ["sphx_api.hip", "sphx_grid.hip", "sphx_knn.hip", "sphx_sums.hip", "sphx_integrate.hip",
           "sphx_loopforms.hip", "sphx_dev.hip", "sphx_refresh.hip", "sphx_blob.hip", "sphx_agb.hip", "sphx_gravity.hip"]
HEADERS = [os.path.join(CSRC, "sphx_internal.h"), os.path.join(CSRC, "sphx_wave.h"),
           os.path.join(os.path.dirname(HERE), "include", "sphx.h")]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsphx.so cannot be built (set HIPCC=/path/to/hipcc)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into sph-code_amd/libsphx.so.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("SPHX_EXTRA_FLAGS", "").split()
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + extra + \
          ["-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
