"""ctypes binding of libtgp.so (include/tgp.h).  Thin plumbing only: no arithmetic here.

The library is the product; there is no CPU fallback.  ``get_ctx()`` raises RuntimeError when
the shared object is missing or no HIP device is present.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGP_LIB_PATH") or os.path.join(_HERE, "csrc", "libtgp.so")   # override: A/B builds

TGP_RBF, TGP_ARBF, TGP_VK, TGP_AVK = 0, 1, 2, 3
NTIMINGS = 11


class TgpKernel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("amp", C.c_double), ("a", C.c_double),
                ("b", C.c_double), ("c", C.c_double), ("ell", C.c_double)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); must list every function declared in include/tgp.h
SIGNATURES = {
    "tgp_init": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "tgp_destroy": (None, [_vp]),
    "tgp_release_caches": (C.c_int, [_vp]),
    "tgp_last_error": (C.c_char_p, [_vp]),
    "tgp_version": (C.c_char_p, []),
    "tgp_device_count": (C.c_int, []),
    "tgp_last_timings": (C.c_int, [_vp, _dp, C.c_int]),
    "tgp_set_profiling": (C.c_int, [_vp, C.c_int]),
    "tgp_set_lookahead": (C.c_int, [_vp, C.c_int]),
    "tgp_kernel_matrix": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _i64, _vp]),
    "tgp_gp_solve": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _vp, _dp, _dp, C.POINTER(_vp)]),
    "tgp_factor_free": (None, [_vp, _vp]),
    "tgp_factor_release": (None, [_vp, _vp]),
    "tgp_factor_borrow": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(_vp)]),
    "tgp_gp_predict": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _i64, _vp]),
    "tgp_gp_predict_cov": (C.c_int, [_vp, _vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _i64, _vp]),
    "tgp_kk_twod": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]),
    "tgp_kk_log": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "tgp_kk_partial": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _i64, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _vp]),
    "tgp_kk_twod_bootstrap": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, C.c_double, C.c_double, C.c_int, _vp]),
    "tgp_vcorr": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, C.c_int, _vp]),
    "tgp_knn_mean": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, C.c_int, _vp]),
    "tgp_binned_stat_2d": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "tgp_dev_alloc": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "tgp_dev_free": (C.c_int, [_vp, _vp]),
    "tgp_h2d": (C.c_int, [_vp, _vp, _vp, _i64]),
    "tgp_d2h": (C.c_int, [_vp, _vp, _vp, _i64]),
    "tgp_sync": (C.c_int, [_vp]),
    "tgp_stream": (_vp, [_vp]),
    "tgp_mem_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "tgp_d_gp_solve": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _vp, _dp, _dp, C.POINTER(_vp)]),
    "tgp_d_gp_predict": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _i64, _vp]),
    "tgp_panel_off": (_i64, [_i64, _i64]),
    "tgp_panel_elems": (_i64, [_i64]),
    "tgp_padded_n": (_i64, [_i64]),
    "tgp_d_kbuild_lower": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp]),
    "tgp_d_potrf": (C.c_int, [_vp, _vp, _i64, _vp]),
    "tgp_d_potrs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "tgp_d_potrs_multi": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.c_int]),
    "tgp_gp_solve_dense": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _dp, _dp, C.POINTER(_vp)]),
    "tgp_d_gp_solve_dense": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _dp, _dp, C.POINTER(_vp)]),
    "tgp_factor_solve": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "tgp_gp_predict_cov_dense": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "tgp_gp_loglik_grad": (C.c_int, [_vp, _vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp]),
    "tgp_d_gp_solve_grad": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _dp, _dp, _vp]),
    "tgp_d_unpack_lower": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "tgp_debug_syrk_loop": (C.c_int, [_vp, _vp, _i64, C.c_int, _dp, _dp]),
    "tgp_debug_tilemap": (C.c_int, [_i64, _vp, _vp, _i64]),
    # multi-GPU tier
    "tgp_set_stream": (C.c_int, [_vp, _vp]),
    "tgp_reset_stream": (C.c_int, [_vp]),
    "tgp_set_side_stream": (C.c_int, [_vp, _vp]),
    "tgp_dist_panel_rows": (_i64, [_i64, _i64, C.c_int, C.c_int]),
    "tgp_dist_panel_off": (_i64, [_i64, _i64, C.c_int, C.c_int]),
    "tgp_dist_local_elems": (_i64, [_i64, C.c_int, C.c_int]),
    "tgp_dd_kbuild": (C.c_int, [_vp, C.POINTER(TgpKernel), _vp, _i64, _vp, _vp, _vp, C.c_int, C.c_int]),
    "tgp_dd_factor_diag": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "tgp_dd_trsm": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "tgp_dd_update": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "tgp_dd_update2": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "tgp_dd_update_group": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "tgp_dd_strip_left": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "tgp_dd_update_group_queued": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "tgp_dd_update_group_fused": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "tgp_dd_wait_head": (C.c_int, [_vp, _vp]),
    "tgp_handoff_mode": (C.c_int, [_vp]),
    "tgp_dd_tail_assemble": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp]),
    "tgp_dd_tail_scatter": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "tgp_dd_queue_reset": (C.c_int, [_vp]),
    "tgp_dd_set_exclusive": (C.c_int, [_vp, C.c_int]),
    "tgp_dd_keep_panel": (C.c_int, [_vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "tgp_dd_fwd_diag": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "tgp_dd_fwd_update": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "tgp_dd_bwd_partial": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "tgp_dd_bwd_diag": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "tgp_dd_logdet_local": (C.c_int, [_vp, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp]),
    "tgp_dd_info": (C.c_int, [_vp, C.c_int]),
}

_lib = None
_ctx = {}
_lock = threading.Lock()


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process do not
    coexist (the second one finds no device), so when torch is installed but not imported yet,
    map its copy first; libtgp.so then binds to it and a later `import torch` (multi-GPU driver)
    shares the same runtime.  No torch code runs here."""
    import sys
    if "torch" in sys.modules or os.environ.get("TGP_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            if os.environ.get("TGP_VERBOSE"):
                print("treegp_amd: using torch's HIP runtime %s (one HIP runtime per process)" % cand, file=sys.stderr)
    except Exception:
        pass


def load_library():
    """dlopen libtgp.so and set the prototypes.  Works without a GPU (no HIP call is made)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtgp.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C treegp_amd/csrc`" % LIB_PATH)
        _share_hip_runtime_with_torch()
        # Contexts that run side by side (the concurrent likelihood evaluations of the ML fit) are limited by the HIP
        # runtime's hardware queues: 4 by default, streams beyond that share queues and serialise (4 contexts at
        # N = 1024: 0.32 ms per evaluation overall with 4 queues, 0.18 ms with 8).  Only effective if HIP has not been
        # initialised in this process yet; an explicit setting wins.
        # Process-wide side effect (documented in INTEGRATION.md); TGP_SET_HW_QUEUES=0 leaves the environment alone.
        if os.environ.get("TGP_SET_HW_QUEUES", "1") != "0":
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class TgpError(RuntimeError):
    pass


_thread = threading.local()


def set_thread_ctx(ctx):
    """This thread's default context (None = back to the process-wide one).  A context serves one caller at a time, so
    threads that compute side by side -- the virtual ranks of the multi-GPU tests -- each bring their own."""
    _thread.ctx = ctx


def release_process_caches(device):
    """the process-wide context of `device`, if one exists, gives back its factor cache / slabs / scratch (tgp_release_caches)"""
    lib = load_library()
    with _lock:                     # (virtual ranks are threads of one process and all come here at once: one at a time)
        ctx = _ctx.get(int(device))
        if ctx is not None:
            lib.tgp_release_caches(ctx)


def get_ctx(device=None):
    """One context per (process, device).  Raises if there is no HIP device -- by design."""
    if device is None and getattr(_thread, "ctx", None) is not None:
        return _thread.ctx
    lib = load_library()
    if device is None:
        device = int(os.environ.get("TGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        if device not in _ctx:
            if lib.tgp_device_count() <= 0:
                raise RuntimeError("treegp_amd needs an AMD GPU (HIP device); none found and there is no CPU path")
            h = _vp()
            dev = (C.c_int * 1)(device)
            rc = lib.tgp_init(dev, 1, C.byref(h))
            if rc != 0:
                raise RuntimeError("tgp_init(device=%d) failed with code %d" % (device, rc))
            _ctx[device] = h
        return _ctx[device]


def new_ctx(device=0):
    """An additional, uncached context on `device` (each context serves one caller at a time)."""
    lib = load_library()
    if lib.tgp_device_count() <= 0:
        raise RuntimeError("treegp_amd needs an AMD GPU (HIP device); none found and there is no CPU path")
    h = _vp()
    dev = (C.c_int * 1)(device)
    rc = lib.tgp_init(dev, 1, C.byref(h))
    if rc != 0:
        raise RuntimeError("tgp_init(device=%d) failed with code %d" % (device, rc))
    return h


class OwnedCtx(object):
    """An additional context that is destroyed (tgp_destroy) with the last reference to this object -- for the multi-GPU
    engine, whose contexts must outlive any factor handle created on them."""

    def __init__(self, device=0):
        self.handle = new_ctx(device)

    def __del__(self):
        try:
            if self.handle is not None and _lib is not None:
                _lib.tgp_destroy(self.handle)
        except Exception:                 # interpreter shutdown
            pass
        self.handle = None


def check(ctx, rc, what):
    """rc < 0 -> TgpError with the library's message; rc > 0 is returned to the caller."""
    if rc < 0:
        msg = load_library().tgp_last_error(ctx)
        err = TgpError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
        err.rc = rc                    # -1: bad argument; -2: a HIP call failed (out of memory, device error, ...)
        raise err
    return rc


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def as_xy(X):
    """(n, ndim in {1,2}) -> contiguous (n, 2) float64, zero second column for 1-D."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    if X.shape[1] == 1:
        X = np.hstack([X, np.zeros_like(X)])
    elif X.shape[1] != 2:
        raise ValueError("only 1-D and 2-D coordinates are supported, got ndim=%d" % X.shape[1])
    return np.ascontiguousarray(X)


def timings(ctx):
    buf = (C.c_double * NTIMINGS)()
    load_library().tgp_last_timings(ctx, buf, NTIMINGS)
    return list(buf)
