"""ctypes binding of libsphx.so (C ABI: include/sphx.h).  No CPU fallback."""
import ctypes as C
import os
import threading
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPHX_LIB") or os.path.join(HERE, "libsphx.so")     # (SPHX_LIB: another build of the same library)

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class SphxConstants(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("k_B", "amu", "m_h", "m_0", "dt_0", "max_age", "pos_clamp")]


class SphxStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_grid", "ms_search", "ms_prep", "ms_density", "ms_pi",
                                          "ms_visc", "ms_integrate", "ms_total")] + \
               [(n, C.c_int64) for n in ("n", "steps", "candidates", "retries", "cells", "refresh_steps",
                                         "rebuild_steps")] + \
               [("cell_size", C.c_double), ("ms_gravity", C.c_double), ("fallback_queries", C.c_int64),
                ("ms_species", C.c_double), ("short_rows", C.c_int64), ("detail_steps", C.c_int64), ("far_queries", C.c_int64),
                ("outlier_levels", C.c_int64), ("bad_accel", C.c_int64), ("bad_energy", C.c_int64), ("bad_state", C.c_int64),
                ("bad_h", C.c_int64), ("graph_steps", C.c_int64), ("search_steps", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# name -> (restype, argtypes); every symbol include/sphx.h declares
SPHX_E_STATE = -4                # include/sphx.h
_P = C.c_void_p
_D = c_double_p
_I = c_int64_p
SIGNATURES = {
    "sphx_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "sphx_destroy": (None, [_P]),
    "sphx_last_error": (C.c_char_p, [_P]),
    "sphx_version": (C.c_int, []),
    "sphx_build_info": (C.c_char_p, []),
    "sphx_tunables": (C.c_char_p, [_P]),
    "sphx_selftest_scan": (C.c_int, [_P, C.c_int, C.c_uint, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "sphx_selftest_mfma_cull": (C.c_int, [_P, C.c_double, C.c_int, C.c_uint, _D, _D, _D]),
    "sphx_host_alloc": (C.c_int, [C.POINTER(_P), C.c_size_t]),
    "sphx_host_free": (C.c_int, [_P]),
    "sphx_set_constants": (C.c_int, [_P, C.POINTER(SphxConstants)]),
    "sphx_get_constants": (C.c_int, [_P, C.POINTER(SphxConstants)]),
    "sphx_set_tuning": (C.c_int, [_P, C.c_double, C.c_double]),
    "sphx_set_incremental": (C.c_int, [_P, C.c_int, C.c_double]),
    "sphx_neighbors": (C.c_int, [_P, C.c_int64, C.c_int, _D, C.c_double, C.c_double, _I, _D, _I, _D]),
    "sphx_hydro_update": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _I] + [_D] * 9 + [C.c_int] + [_D] * 7),
    "sphx_density": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D, _D, _I, C.c_double, _D]),
    "sphx_dust_density": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D, _I, _D, _D, _D]),
    "sphx_num_dens": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D, _D, _I, C.c_double, _D]),
    "sphx_del_pressure": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D, _D, _I, _D, _D, C.c_double, _D]),
    "sphx_artificial_viscosity": (C.c_int, [_P, C.c_int64, C.c_int, _I] + [_D] * 9 + [C.c_double, _D, _D]),
    "sphx_crossing_time": (C.c_int, [_P, C.c_int64, C.c_int, _I, _D, _D, _D, _D]),
    "sphx_net_impulse": (C.c_int, [_P, C.c_int64, C.c_int, _D, _D, _D, _D, _D, _I, _D, _D, _D, _D]),
    "sphx_agb_yields": (C.c_int, [_P, C.c_int64, _D, _D, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _D, _D, _D,
                                  C.POINTER(C.c_int32), C.c_double, C.c_int, _D, _D, C.c_double, _D, _D]),
    "sphx_dt_rule": (C.c_int, [_P, C.c_int64, _D, C.POINTER(C.c_int32), _D]),
    "sphx_clamp_arrays": (C.c_int, [_P, C.c_int64, _D, _D]),
    "sphx_leapfrog": (C.c_int, [_P, C.c_int64] + [_D] * 18 + [C.c_double]),
    "sphx_state_upload": (C.c_int, [_P, C.c_int64, C.c_int] + [_D] * 10),
    "sphx_state_set_drag": (C.c_int, [_P, _D, _D]),
    "sphx_state_set_agb": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _D, _D, _D,
                                     C.POINTER(C.c_int32), C.c_double, _D, C.c_double]),
    "sphx_state_download_species": (C.c_int, [_P, _D, _D, _D]),
    "sphx_state_set_loop_forms": (C.c_int, [_P, C.c_int, C.c_double]),
    "sphx_set_clip_grad": (C.c_int, [_P, C.c_int]),
    "sphx_state_set_gravity": (C.c_int, [_P, C.c_int, C.c_double]),
    "sphx_set_gravity_order": (C.c_int, [_P, C.c_int]),
    "sphx_gravity_direct": (C.c_int, [_P, C.c_int64, _D, _D, _D, C.c_double, C.c_double, _D]),
    "sphx_gravity_tree": (C.c_int, [_P, C.c_int64, _D, _D, _D, C.c_double, C.c_double, C.c_int, C.c_int, _D]),
    "sphx_step": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]),
    "sphx_state_download": (C.c_int, [_P] + [_D] * 10),
    "sphx_state_download_pressure": (C.c_int, [_P, _D]),
    "sphx_get_stats": (C.c_int, [_P, C.POINTER(SphxStats)]),
    "sphx_reset_stats": (C.c_int, [_P]),
    "sphx_set_stream": (C.c_int, [_P, C.c_void_p]),
    "sphx_reset_stream": (C.c_int, [_P]),
    "sphx_sync": (C.c_int, [_P]),
    "sphx_dev_set_mean_h": (C.c_int, [_P, C.c_double]),
    "sphx_dev_search": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_double, C.c_double, _P]),
    "sphx_dev_get_order": (C.c_int, [_P, C.c_int64, _P]),
    "sphx_dev_prep": (C.c_int, [_P] + [_P] * 8),
    "sphx_dev_density": (C.c_int, [_P] + [_P] * 4),
    "sphx_dev_pi": (C.c_int, [_P] + [_P] * 4),
    "sphx_dev_visc": (C.c_int, [_P] + [_P] * 4),
    "sphx_dev_clamp": (C.c_int, [_P, C.c_int64, _P, _P]),
    "sphx_dev_pack_rows": (C.c_int, [_P, C.c_int64, _P, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P]),
    "sphx_dev_regroup": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "sphx_dev_need_map": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(C.c_double), C.c_double, C.c_int, _P]),
    "sphx_dev_integrate_auto": (C.c_int, [_P, C.c_int64] + [_P] * 13 + [C.c_int, C.c_double, _P]),
    "sphx_dev_reach": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.c_double, C.c_double, _P]),
    "sphx_set_timing_detail": (C.c_int, [_P, C.c_int]),
    "sphx_dev_reach_dt": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.c_double, _P, _P]),
    "sphx_dev_plan_mask": (C.c_int, [_P, C.c_int64, _P, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "sphx_dev_step_scalars": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.c_double, _P, _P]),
    "sphx_dev_integrate": (C.c_int, [_P, C.c_int64] + [_P] * 12 + [C.c_double]),
    "sphx_dev_drag": (C.c_int, [_P] + [_P] * 6),
    "sphx_dev_set_drag_terms": (C.c_int, [_P] + [_P] * 4),
    "sphx_dev_set_agb": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _D, _D, _D,
                                   C.POINTER(C.c_int32), C.c_double, _D, C.c_double]),
    "sphx_dev_species": (C.c_int, [_P, C.c_int] + [_P] * 5),
    "sphx_dev_loop_prep": (C.c_int, [_P] + [_P] * 8 + [C.c_double]),
    "sphx_dev_loop_pass1": (C.c_int, [_P] + [_P] * 5),
    "sphx_dev_loop_pass2": (C.c_int, [_P] + [_P] * 4),
    "sphx_dev_loop_pass2_interior": (C.c_int, [_P]),
    "sphx_dev_set_reach_cap": (C.c_int, [_P, C.c_double]),
    "sphx_dev_select_blobs": (C.c_int, [_P, C.c_int]),
    "sphx_dev_blob_split_counts": (C.c_int, [_P, _P]),
    "sphx_dev_integrate_loop": (C.c_int, [_P, C.c_int64] + [_P] * 14 + [C.c_int, C.c_double, C.c_double, _P]),
}

_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen libsphx.so and attach prototypes.  Raises if the library is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsphx.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python sph-code_amd/build.py`. There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def f64(a, shape=None):
    """C-contiguous float64 view/copy (mass and f_un arrive as longdouble from the driver,
    sph/code_running.py:152-154: cast, as nsc:559-565 does for the other arrays)."""
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != shape:
        raise ValueError("expected shape %s, got %s" % (shape, out.shape))
    return out


def i64(a, shape=None):
    out = np.ascontiguousarray(a, dtype=np.int64)
    if shape is not None and out.shape != shape:
        raise ValueError("expected shape %s, got %s" % (shape, out.shape))
    return out


class _PinnedPool:
    """Page-locked host buffers for large result arrays (include/sphx.h sphx_host_alloc).  A buffer backs a
    NumPy array; when the last view of it dies the buffer returns to the pool (locking pages costs more than
    the copy it speeds up, so buffers are reused, a few per size).  Falls back to ordinary memory when pages
    cannot be locked."""
    MIN_BYTES = 8 << 20
    KEEP_PER_SIZE = 3

    def __init__(self):
        self.free = {}
        self.lock = threading.Lock()

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        with self.lock:
            lst = self.free.get(nbytes)
            ptr = lst.pop() if lst else None
        if ptr is None:
            lib = load_library()
            p = _P()
            if lib.sphx_host_alloc(C.byref(p), nbytes) != 0 or not p.value:
                return np.empty(shape, dtype)
            ptr = p.value
        buf = (C.c_byte * nbytes).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, nbytes)       # (runs when the last array over buf is gone)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def _release(self, ptr, nbytes):
        with self.lock:
            lst = self.free.setdefault(nbytes, [])
            if len(lst) < self.KEEP_PER_SIZE:
                lst.append(ptr)
                return
        try:
            load_library().sphx_host_free(_P(ptr))
        except Exception:
            pass


pinned = _PinnedPool()


def dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def ip(a):
    return None if a is None else a.ctypes.data_as(c_int64_p)


class Context:
    """One libsphx context = one GPU + one HIP stream.  Not thread-safe."""

    def __init__(self, device=None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        h = C.c_void_p()
        rc = self.lib.sphx_create(C.byref(h), device)
        if rc != 0 or not h:
            raise RuntimeError("sphx_create(device=%d) failed with %d: no usable MI355X/HIP device "
                               "(libsphx has no CPU path)" % (device, rc))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sphx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == 0:
            return
        msg = self.lib.sphx_last_error(self.h)
        msg = msg.decode() if msg else ""
        if rc == -1:
            raise ValueError("libsphx: " + msg)
        raise RuntimeError("libsphx error %d: %s" % (rc, msg))

    def constants(self):
        c = SphxConstants()
        self.check(self.lib.sphx_get_constants(self.h, C.byref(c)))
        return c

    def set_constants(self, **kw):
        c = self.constants()
        for k_, v in kw.items():
            setattr(c, k_, v)
        self.check(self.lib.sphx_set_constants(self.h, C.byref(c)))

    def set_tuning(self, rscale=0.0, cell_factor=0.0):
        self.check(self.lib.sphx_set_tuning(self.h, float(rscale), float(cell_factor)))

    def set_incremental(self, verlet=True, rscale_build=0.0):
        self.check(self.lib.sphx_set_incremental(self.h, 1 if verlet else 0, float(rscale_build)))

    def stats(self):
        s = SphxStats()
        self.check(self.lib.sphx_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self.check(self.lib.sphx_reset_stats(self.h))

    def build_info(self):
        """-> dict: how libsphx.so was built and which SPHX_* tunables this context read from the environment."""
        return {"library": self.lib.sphx_build_info().decode(), "tunables": self.lib.sphx_tunables(self.h).decode(),
                "lib_path": LIB_PATH}

    def set_timing_detail(self, on=True):
        """Per-pass timing events in sphx_step (ms_prep ... ms_integrate); off by default: each costs the stream ~10 us."""
        self.check(self.lib.sphx_set_timing_detail(self.h, 1 if on else 0))


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context()
    return _default
