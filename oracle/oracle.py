"""ctypes front-end of the CPU oracle (oracle/dflow_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.  Arrays use the reference's own dtypes (daisy i flann.py:89-95, Q14): proposals int64
(H,W,L,2) [-1 fill], lcosts float64 (H,W,L) [1000.0 fill], nprop/bestlabels int64 (H,W), flow float64
(H,W,2) [dy,dx].
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    """Mirror of orc_params.  Defaults = reference constants (daisy i flann.py:34-48,88,172,207-208)."""
    _fields_ = [("pich", C.c_int32), ("picw", C.c_int32), ("cellh", C.c_int32), ("cellw", C.c_int32),
                ("maxnprop", C.c_int32), ("knn", C.c_int32), ("window", C.c_int32), ("ngauss", C.c_int32),
                ("tpsi", C.c_int32), ("max_attempts", C.c_int32), ("tphi", C.c_float), ("sigma", C.c_float),
                ("lamda", C.c_double), ("seed", C.c_uint64)]


def make_params(pich, picw, cellh, cellw, seed=0, **kw):
    p = Params(pich=pich, picw=picw, cellh=cellh, cellw=cellw, maxnprop=150, knn=5, window=2, ngauss=25,
               tpsi=8, max_attempts=1 << 16, tphi=2.5, sigma=8.0, lamda=0.05, seed=seed)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_gauss_offset.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    """Threads for the loops over independent units (rows / columns / chains); results do not depend on n.
    The reference is single-threaded: the default is 1."""
    lib().orc_set_threads(C.c_int(int(n)))


def get_threads():
    return int(lib().orc_get_threads())


def daisy(bgr):
    """izracunajDaisy (daisy i flann.py:69-77): (H,W,3) uint8 BGR -> (H,W,68) float32."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    H, W, _ = bgr.shape
    out = np.empty((H, W, 68), np.float32)
    lib().orc_daisy(_p(bgr), C.c_int(H), C.c_int(W), _p(out))
    return out


def daisy_cubes(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    H, W, _ = bgr.shape
    out = np.empty((4, H, W, 4), np.float32)
    lib().orc_daisy_cubes(_p(bgr), C.c_int(H), C.c_int(W), _p(out))
    return out


def knn_cell(p, q, d2, ci, cj):
    """Canonical exact 5-NN of one query descriptor in cell (ci,cj): (idx int32[K], dist float32[K])."""
    q = np.ascontiguousarray(q, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    idx = np.zeros(16, np.int32)
    dist = np.zeros(16, np.float32)
    lib().orc_knn_cell(C.byref(p), _p(q), _p(d2), C.c_int(ci), C.c_int(cj), _p(idx), _p(dist))
    return idx[:p.knn].copy(), dist[:p.knn].copy()


def knn_proposals(p, d1, d2):
    """generisi (daisy i flann.py:157-189) with the canonical exact search."""
    H, W, L = p.pich, p.picw, p.maxnprop
    d1 = np.ascontiguousarray(d1, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    proposals = np.empty((H, W, L, 2), np.int64)
    lcosts = np.empty((H, W, L), np.float64)
    nprop = np.empty((H, W), np.int64)
    bestlabels = np.empty((H, W), np.int64)
    lib().orc_knn_proposals(C.byref(p), _p(d1), _p(d2), _p(proposals), _p(lcosts), _p(nprop), _p(bestlabels))
    return proposals, lcosts, nprop, bestlabels


def neighbour_proposals(p, d1, d2, proposals, lcosts, nprop, bestlabels, want_attempts=False):
    """nasumicni (daisy i flann.py:205-233); proposals/lcosts/nprop are updated in place."""
    for a, dt in ((proposals, np.int64), (lcosts, np.float64), (nprop, np.int64), (bestlabels, np.int64)):
        assert a.dtype == dt and a.flags.c_contiguous
    d1 = np.ascontiguousarray(d1, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    att = np.zeros((p.pich, p.picw), np.int32) if want_attempts else None
    lib().orc_neighbour_proposals(C.byref(p), _p(d1), _p(d2), _p(proposals), _p(lcosts), _p(nprop), _p(bestlabels),
                                  _p(att) if want_attempts else None)
    return att


def pack_compat(p, proposals, nprop):
    """pakovanje (daisy i flann.py:256-309): (H,W,2,L*L//8+1) uint8."""
    L = p.maxnprop
    packed = np.zeros((p.pich, p.picw, 2, L * L // 8 + 1), np.uint8)
    lib().orc_pack_compat(C.byref(p), _p(proposals), _p(nprop), _p(packed))
    return packed


def bcd_chain(p, proposals, lcosts, nprop, bestlabels, ystep, xstep, ty, tx):
    """bcd (python bcd.py:101-257); bestlabels updated in place."""
    lib().orc_bcd_chain(C.byref(p), _p(proposals), _p(lcosts), _p(nprop), _p(bestlabels),
                        C.c_int(ystep), C.c_int(xstep), C.c_int(ty), C.c_int(tx))


def bcd_phase(p, proposals, lcosts, nprop, bestlabels, phase):
    lib().orc_bcd_phase(C.byref(p), _p(proposals), _p(lcosts), _p(nprop), _p(bestlabels), C.c_int(phase))


def bcd_sweep(p, proposals, lcosts, nprop, bestlabels):
    """One iteration of ceoBCD's loop body (python bcd.py:265-277)."""
    lib().orc_bcd_sweep(C.byref(p), _p(proposals), _p(lcosts), _p(nprop), _p(bestlabels))


def labels_to_flow(p, proposals, bestlabels):
    """vratiKonacniFlow (python bcd.py:90-95): (H,W,2) float64 [dy,dx]."""
    flow = np.empty((p.pich, p.picw, 2), np.float64)
    lib().orc_labels_to_flow(C.byref(p), _p(proposals), _p(bestlabels), _p(flow))
    return flow


def fb_consistency(fwd, bwd, tresh):
    """postProcessing (postprocessing.py:123-135) on two (H,W,2) [dy,dx] fields -> (H,W,3) float32 [U,V,valid]."""
    fwd = np.ascontiguousarray(fwd, np.float64)
    bwd = np.ascontiguousarray(bwd, np.float64)
    H, W, _ = fwd.shape
    out = np.empty((H, W, 3), np.float32)
    lib().orc_fb_consistency(C.c_int(H), C.c_int(W), _p(fwd), _p(bwd), C.c_double(tresh), _p(out))
    return out


def gauss_thresholds(sigma):
    thr = np.empty(127, np.uint32)
    lib().orc_gauss_thresholds(C.c_double(sigma), _p(thr))
    return thr


def gauss_offset(thr, u):
    return int(lib().orc_gauss_offset(_p(thr), C.c_uint32(u)))


def philox(c0, c1, seed):
    out = np.empty(4, np.uint32)
    lib().orc_philox(C.c_uint32(c0), C.c_uint32(c1), C.c_uint64(seed), _p(out))
    return out


def full_pass(p, img1, img2, bcd_times):
    """daisy i flann.py main (:406-422) + ceoBCD (python bcd.py:261-284) on one image pair."""
    d1, d2 = daisy(img1), daisy(img2)
    proposals, lcosts, nprop, bestlabels = knn_proposals(p, d1, d2)
    flow0 = labels_to_flow(p, proposals, bestlabels)
    neighbour_proposals(p, d1, d2, proposals, lcosts, nprop, bestlabels)
    flows = [flow0]
    for _ in range(bcd_times):
        bcd_sweep(p, proposals, lcosts, nprop, bestlabels)
        flows.append(labels_to_flow(p, proposals, bestlabels))
    return dict(d1=d1, d2=d2, proposals=proposals, lcosts=lcosts, nprop=nprop, bestlabels=bestlabels, flows=flows)


def knn_points(q, pts, K=5):
    """Canonical exact K-NN of q among the rows of pts (stand-in for flann.nn_index, daisy i flann.py:171)."""
    q = np.ascontiguousarray(q, np.float32)
    pts = np.ascontiguousarray(pts, np.float32)
    idx = np.zeros(16, np.int32)
    dist = np.zeros(16, np.float32)
    lib().orc_knn_points(_p(q), _p(pts), C.c_int(pts.shape[0]), C.c_int(K), _p(idx), _p(dist))
    return idx[:K].copy(), dist[:K].copy()
