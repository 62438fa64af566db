"""ctypes binding of libdflow.so (include/dflow.h).  There is no CPU fallback: if the HIP library is missing
or a call fails, an exception is raised."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdflow.so")

SYMBOLS = ("dflow_version", "dflow_last_error", "dflow_default_params", "dflow_workspace_bytes", "dflow_daisy",
           "dflow_knn_proposals", "dflow_knn_proposals_timed", "dflow_knn_screen_stats", "dflow_neighbour_proposals", "dflow_bcd_prepare", "dflow_bcd_phase", "dflow_bcd_sweep",
           "dflow_bcd_phase_batch", "dflow_bcd_sweep_batch",
           "dflow_labels_to_flow", "dflow_fb_consistency", "dflow_pack_compat", "dflow_remove_small_segments_host")


FLAG_KNN_EXACT = 1      # DFLOW_FLAG_KNN_EXACT
FLAG_DESCR_F16 = 8      # DFLOW_FLAG_DESCR_F16
DESC_PITCH_F16 = 72     # DFLOW_DESC_PITCH_F16: binary16 descriptor planes are (H,W,72)


class DflowError(RuntimeError):
    pass


class Params(C.Structure):
    """struct dflow_params (include/dflow.h) = the module globals of daisy i flann.py:34-48,88,172,207-208."""
    _fields_ = [("pich", C.c_int32), ("picw", C.c_int32), ("cellh", C.c_int32), ("cellw", C.c_int32),
                ("maxnprop", C.c_int32), ("knn", C.c_int32), ("window", C.c_int32), ("ngauss", C.c_int32),
                ("tpsi", C.c_int32), ("max_attempts", C.c_int32), ("tphi", C.c_float), ("sigma", C.c_float),
                ("lamda", C.c_double), ("seed", C.c_uint64), ("label_pitch", C.c_int32), ("flags", C.c_int32)]


def build(force=False):
    """Compile libdflow.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "libdflow.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DflowError("%s is missing: run __graft_entry__.build() (there is no CPU fallback)" % LIB_PATH)
        # torch brings its own copy of the HIP runtime (same SONAME as /opt/rocm's).  If libdflow.so is loaded first it pulls in
        # the system copy, torch then loads its own, and whichever initialises second finds "no ROCm-capable device"
        # (python __graft_entry__.py smoke = build() then smoke() in one process did exactly that).  torch first: one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
        pp = C.POINTER(Params)
        L.dflow_version.restype = C.c_int
        L.dflow_last_error.restype = C.c_char_p
        L.dflow_default_params.argtypes = [pp, i32, i32, i32, i32]
        L.dflow_default_params.restype = None
        L.dflow_workspace_bytes.argtypes = [pp]
        L.dflow_workspace_bytes.restype = sz
        L.dflow_daisy.argtypes = [pp, vp, vp, vp, sz, vp]
        L.dflow_knn_proposals.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.dflow_knn_proposals_timed.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.dflow_knn_screen_stats.argtypes = [pp, vp, sz, vp, C.POINTER(C.c_int64)]
        L.dflow_neighbour_proposals.argtypes = [pp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
        L.dflow_bcd_prepare.argtypes = [pp, vp, vp, vp, vp, sz, vp]
        L.dflow_bcd_phase.argtypes = [pp, vp, vp, vp, i32, vp, sz, vp]
        L.dflow_bcd_sweep.argtypes = [pp, vp, vp, vp, vp, sz, vp]
        L.dflow_bcd_phase_batch.argtypes = [pp, i32, vp, vp, i32, vp, sz, vp]
        L.dflow_bcd_sweep_batch.argtypes = [pp, i32, vp, vp, vp, sz, vp]
        L.dflow_labels_to_flow.argtypes = [pp, vp, vp, vp, vp]
        L.dflow_fb_consistency.argtypes = [pp, vp, vp, C.c_float, vp, vp]
        L.dflow_pack_compat.argtypes = [pp, vp, vp, vp, vp]
        L.dflow_remove_small_segments_host.argtypes = [vp, i32, i32, C.c_float, i32]
        for n in SYMBOLS[4:]:
            getattr(L, n).restype = C.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise DflowError("%s failed (%d): %s" % (what, rc, lib().dflow_last_error().decode()))


def default_params(pich, picw, cellh, cellw, **kw):
    p = Params()
    lib().dflow_default_params(C.byref(p), pich, picw, cellh, cellw)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p
