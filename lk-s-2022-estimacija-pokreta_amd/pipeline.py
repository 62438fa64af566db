"""Host-side mirror of the reference's hot path on one MI355X.

`DiscreteFlow` keeps the state the reference keeps in module globals (daisy i flann.py:89-95) as device
tensors and exposes the reference's own function names; each method is one call through the C-ABI
(include/dflow.h) on torch's current HIP stream.  torch is used for device memory, streams and
torch.distributed only -- every computation happens in libdflow.so.  No CPU fallback exists.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import flowio

DEFAULT_CELLS = {  # (pich, picw) -> (cellh, cellw)
    (375, 1241): (25, 73),    # daisy i flann.py:34-35,42-43 (KITTI, exact tiling)
    (375, 1242): (25, 54),    # discrete_flow.py:22-23,30-31
    (436, 1024): (27, 64),    # Sintel: 16x16 cells, last cell row absorbs 4 rows
}


def default_cells(pich, picw):
    if (pich, picw) in DEFAULT_CELLS:
        return DEFAULT_CELLS[(pich, picw)]
    return max(5, pich // 15), max(5, picw // 17)   # the reference's 17x15 grid (daisy i flann.py:85-86)


class DiscreteFlow:
    """One (pair, direction) pass: DAISY -> kNN proposals -> neighbour proposals -> BCD sweeps."""

    def __init__(self, pich, picw, cellh=None, cellw=None, device="cuda:0", seed=0, **overrides):
        if not torch.cuda.is_available():
            raise _lib.DflowError("no HIP device visible: the dflow hot path has no CPU fallback")
        if cellh is None or cellw is None:
            cellh, cellw = default_cells(pich, picw)
        self.p = _lib.default_params(pich, picw, cellh, cellw, seed=seed, **overrides)
        _lib.check(0 if _lib.lib().dflow_workspace_bytes(C.byref(self.p)) else -1, "dflow_workspace_bytes")
        self.device = torch.device(device)
        self._descr_f16 = bool(self.p.flags & _lib.FLAG_DESCR_F16)     # storage mode of the descriptor planes: fixed here
        H, W, LP = pich, picw, self.p.label_pitch
        dev = self.device
        # float32 (H,W,68), or with DFLOW_FLAG_DESCR_F16 binary16 (H,W,72): 68 values + 4 zero pads per pixel (include/dflow.h)
        self.descrs1 = self._new_descr()                                             # daisy i flann.py:80
        self.descrs2 = self._new_descr()                                             # :81
        self.proposals = torch.empty((H, W, LP), dtype=torch.int32, device=dev)     # :89 (packed int16 pairs)
        self.lcosts = torch.empty((H, W, LP), dtype=torch.float32, device=dev)      # :90
        self.nprop = torch.empty((H, W), dtype=torch.int32, device=dev)             # :91
        self.bestlabels = torch.empty((H, W), dtype=torch.int32, device=dev)        # :95
        self.flow = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
        self.ws_bytes = int(_lib.lib().dflow_workspace_bytes(C.byref(self.p)))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self._img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        self._bcd_ready = False     # compat matrices in the workspace are valid for the current proposals

    # ------------------------------------------------------------------ helpers
    @property
    def descr_f16(self):
        return self._descr_f16

    def _new_descr(self):
        H, W = self.p.pich, self.p.picw
        if self.descr_f16:
            return torch.zeros((H, W, _lib.DESC_PITCH_F16), dtype=torch.float16, device=self.device)
        return torch.empty((H, W, 68), dtype=torch.float32, device=self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _pp(self):
        # the planes were allocated for one storage mode; a flag flipped afterwards would make the kernels read or write
        # 272-byte rows in 144-byte rows (or the reverse)
        if bool(self.p.flags & _lib.FLAG_DESCR_F16) != self._descr_f16:
            raise _lib.DflowError("DFLOW_FLAG_DESCR_F16 changed after construction: the descriptor planes are %s"
                                  % ("binary16 (H,W,72)" if self._descr_f16 else "float32 (H,W,68)"))
        return C.byref(self.p)

    # ------------------------------------------------------------------ reference-named stages
    def izracunajDaisy(self, picture, out=None):
        """daisy i flann.py:69-77.  picture: (H,W,3) uint8 BGR (numpy or device tensor) -> (H,W,68) f32 tensor (with
        DFLOW_FLAG_DESCR_F16: (H,W,72) binary16, see descriptors_f32)."""
        H, W = self.p.pich, self.p.picw
        if isinstance(picture, np.ndarray):
            if picture.shape != (H, W, 3) or picture.dtype != np.uint8:
                raise ValueError("picture must be uint8 (%d,%d,3)" % (H, W))
            self._img.copy_(torch.from_numpy(np.ascontiguousarray(picture)))
            img = self._img
        else:
            if tuple(picture.shape) != (H, W, 3) or picture.dtype != torch.uint8 or not picture.is_contiguous():
                raise ValueError("picture must be a contiguous uint8 (%d,%d,3) tensor" % (H, W))
            img = picture
        if out is None:
            out = self._new_descr()
        self._bcd_ready = False
        _lib.check(_lib.lib().dflow_daisy(self._pp(), img.data_ptr(), out.data_ptr(), self.ws.data_ptr(),
                                          self.ws_bytes, self._stream()), "dflow_daisy")
        return out

    def load_pair(self, pic3, pic4):
        """daisy i flann.py:406-407."""
        self.izracunajDaisy(pic3, out=self.descrs1)
        self.izracunajDaisy(pic4, out=self.descrs2)

    def set_descriptors(self, descrs1, descrs2):
        """(H,W,68) arrays -> the descriptor planes (rounded to binary16 if the pass stores them that way)."""
        for dst, src in ((self.descrs1, descrs1), (self.descrs2, descrs2)):
            dst[..., :68].copy_(torch.as_tensor(src, dtype=torch.float32).to(dst.dtype))

    def descriptors_f32(self, which):
        """Descriptors of image `which` (0: first, 1: second) as an (H,W,68) float32 tensor, whatever the storage."""
        return (self.descrs1, self.descrs2)[which][..., :68].to(torch.float32)

    def generisi(self):
        """napraviCD2 + generisi, daisy i flann.py:144-189."""
        self._bcd_ready = False
        _lib.check(_lib.lib().dflow_knn_proposals(self._pp(), self.descrs1.data_ptr(), self.descrs2.data_ptr(),
                                                  self.proposals.data_ptr(), self.lcosts.data_ptr(),
                                                  self.nprop.data_ptr(), self.bestlabels.data_ptr(),
                                                  self.ws.data_ptr(), self.ws_bytes, self._stream()),
                   "dflow_knn_proposals")

    KNN_KERNELS = ("basis", "prep", "knn_screen_kernel", "knn_resolve_kernel", "knn_fix_kernel", "knn_finalize_kernel")

    def generisi_timed(self):
        """generisi with HIP events between its kernels (dflow_knn_proposals_timed): returns ({kernel: ms}, MFMAs issued)."""
        self._bcd_ready = False
        ms = (C.c_float * 6)()
        issued = C.c_double()
        _lib.check(_lib.lib().dflow_knn_proposals_timed(self._pp(), self.descrs1.data_ptr(), self.descrs2.data_ptr(),
                                                        self.proposals.data_ptr(), self.lcosts.data_ptr(),
                                                        self.nprop.data_ptr(), self.bestlabels.data_ptr(),
                                                        self.ws.data_ptr(), self.ws_bytes, self._stream(), ms, C.byref(issued)),
                   "dflow_knn_proposals_timed")
        return dict(zip(self.KNN_KERNELS, (float(v) for v in ms))), float(issued.value)

    KNN_STATS = ("lists_exact", "flags", "lists", "entries", "events", "max_entries_per_lane", "zero_queries", "bad_queries",
                 "zero_candidates", "zero_candidates_removed", "query_cell_pairs", "list_capacity", "heavy_pairs")

    def knn_stats(self):
        """dflow_knn_screen_stats: what the MFMA screen of the last generisi() did (call before the next stage reuses the workspace)."""
        out = (C.c_int64 * len(self.KNN_STATS))()
        _lib.check(_lib.lib().dflow_knn_screen_stats(self._pp(), self.ws.data_ptr(), self.ws_bytes, self._stream(), out), "dflow_knn_screen_stats")
        st = dict(zip(self.KNN_STATS, (int(v) for v in out)))
        st["events_per_query_cell"] = round(st["events"] / max(1, st["query_cell_pairs"]), 3)
        return st

    def nasumicni(self):
        """daisy i flann.py:205-233."""
        self._bcd_ready = False
        _lib.check(_lib.lib().dflow_neighbour_proposals(self._pp(), self.descrs1.data_ptr(), self.descrs2.data_ptr(),
                                                        self.proposals.data_ptr(), self.lcosts.data_ptr(),
                                                        self.nprop.data_ptr(), self.bestlabels.data_ptr(),
                                                        self.ws.data_ptr(), self.ws_bytes, self._stream()),
                   "dflow_neighbour_proposals")

    def pakovanje(self):
        """daisy i flann.py:256-309: compat bit matrices, built into the workspace for the chain kernel."""
        _lib.check(_lib.lib().dflow_bcd_prepare(self._pp(), self.proposals.data_ptr(), self.lcosts.data_ptr(),
                                                self.nprop.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self._stream()),
                   "dflow_bcd_prepare")
        self._bcd_ready = True

    def bcd_phase(self, phase):
        """One of the four chain loops of ceoBCD, python bcd.py:265-277."""
        if not self._bcd_ready:
            self.pakovanje()
        _lib.check(_lib.lib().dflow_bcd_phase(self._pp(), self.proposals.data_ptr(), self.nprop.data_ptr(),
                                              self.bestlabels.data_ptr(), phase, self.ws.data_ptr(), self.ws_bytes,
                                              self._stream()), "dflow_bcd_phase")

    def ceoBCD(self, bcd_times, on_sweep=None):
        """python bcd.py:261-284.  on_sweep(w) is called after sweep w (the reference saves .npy there)."""
        if not self._bcd_ready:
            self.pakovanje()
        for w in range(1, bcd_times + 1):
            _lib.check(_lib.lib().dflow_bcd_sweep(self._pp(), self.proposals.data_ptr(), self.nprop.data_ptr(),
                                                  self.bestlabels.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                  self._stream()), "dflow_bcd_sweep")
            if on_sweep is not None:
                on_sweep(w)

    def vratiKonacniFlow(self, out=None):
        """python bcd.py:90-95: (H,W,2) [dy,dx] (float32 device tensor; values are small integers)."""
        out = self.flow if out is None else out
        _lib.check(_lib.lib().dflow_labels_to_flow(self._pp(), self.proposals.data_ptr(), self.bestlabels.data_ptr(),
                                                   out.data_ptr(), self._stream()), "dflow_labels_to_flow")
        return out

    def run(self, pic3, pic4, bcd_times):
        """daisy i flann.py main (:406-422, without the file writes) followed by ceoBCD; returns the flow tensor."""
        self.load_pair(pic3, pic4)
        self.generisi()
        self.nasumicni()
        self.ceoBCD(bcd_times)
        return self.vratiKonacniFlow()

    # ------------------------------------------------------------------ reference dtypes on the host
    def host_state(self):
        """proposals int64 (H,W,150,2) / lcosts float64 / nprop, bestlabels int64, as the reference saves them
        (daisy i flann.py:249-253)."""
        L = self.p.maxnprop
        packed = self.proposals[..., :L].cpu().numpy().view(np.uint32)
        dy = (packed & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.int64)
        dx = (packed >> 16).astype(np.uint16).view(np.int16).astype(np.int64)
        return dict(proposals=np.stack([dy, dx], axis=-1),
                    lcosts=self.lcosts[..., :L].cpu().numpy().astype(np.float64),
                    nprop=self.nprop.cpu().numpy().astype(np.int64),
                    bestlabels=self.bestlabels.cpu().numpy().astype(np.int64))

    def set_host_state(self, proposals, lcosts, nprop, bestlabels):
        """Upload reference-dtype arrays (ucitajSvePodatkeDoBCD, python bcd.py:67-81)."""
        L, LP = self.p.maxnprop, self.p.label_pitch
        H, W = self.p.pich, self.p.picw
        packed = np.full((H, W, LP), 0xFFFFFFFF, np.uint32)
        packed[..., :L] = (proposals[..., 0].astype(np.int16).view(np.uint16).astype(np.uint32)
                           | (proposals[..., 1].astype(np.int16).view(np.uint16).astype(np.uint32) << 16))
        lc = np.full((H, W, LP), 1000.0, np.float32)
        lc[..., :L] = lcosts.astype(np.float32)
        # the device keeps the data costs as float32: exact for everything the reference writes (min(tphi, a float32 sum),
        # daisy i flann.py:179-180,228-229) -- anything else would silently change the DP, so it is refused
        if not np.array_equal(lc[..., :L].astype(np.float64), lcosts):
            raise ValueError("lcosts holds values that are not float32-exact (the reference's files are): refusing to round them")
        if proposals.min() < -32768 or proposals.max() > 32767:
            raise ValueError("proposals outside the int16 range of the packed device layout")
        self.proposals.copy_(torch.from_numpy(packed.view(np.int32)))
        self.lcosts.copy_(torch.from_numpy(lc))
        self.nprop.copy_(torch.from_numpy(nprop.astype(np.int32)))
        self.bestlabels.copy_(torch.from_numpy(bestlabels.astype(np.int32)))
        self._bcd_ready = False


def ceoBCD_batch(passes, bcd_times, on_sweep=None):
    """ceoBCD (python bcd.py:261-284) for several independent passes at once (forward and backward runs of a pair, several
    pairs: README.md:40 of the reference): the chains of all passes share the four launches of a sweep.  `passes` are
    DiscreteFlow objects of identical geometry and constants on one device; results equal separate ceoBCD calls."""
    passes = list(passes)
    if not passes:
        return
    first = passes[0]
    for df in passes:
        if bytes(df.p) != bytes(first.p) and (df.p.pich, df.p.picw, df.p.cellh, df.p.cellw, df.p.tpsi, df.p.lamda, df.p.label_pitch, df.p.maxnprop) != \
                (first.p.pich, first.p.picw, first.p.cellh, first.p.cellw, first.p.tpsi, first.p.lamda, first.p.label_pitch, first.p.maxnprop):
            raise ValueError("batched passes must share geometry and constants")
        if df.device != first.device:
            raise ValueError("batched passes must live on one device")
        if not df._bcd_ready:
            df.pakovanje()
    n = len(passes)
    arr = C.c_void_p * n
    nprop = arr(*[df.nprop.data_ptr() for df in passes])
    best = arr(*[df.bestlabels.data_ptr() for df in passes])
    ws = arr(*[df.ws.data_ptr() for df in passes])
    for w in range(1, bcd_times + 1):
        _lib.check(_lib.lib().dflow_bcd_sweep_batch(first._pp(), n, nprop, best, ws, first.ws_bytes, first._stream()),
                   "dflow_bcd_sweep_batch")
        if on_sweep is not None:
            on_sweep(w)


def fb_consistency(fwd, bwd, tresh, p=None):
    """postProcessing (postprocessing.py:123-135) on two (H,W,2) [dy,dx] float32 device tensors ->
    (H,W,3) float32 [U,V,valid] device tensor."""
    H, W, _ = fwd.shape
    if p is None:
        ch, cw = default_cells(H, W)
        p = _lib.default_params(H, W, ch, cw)
    out = torch.empty((H, W, 3), dtype=torch.float32, device=fwd.device)
    stream = C.c_void_p(torch.cuda.current_stream(fwd.device).cuda_stream)
    _lib.check(_lib.lib().dflow_fb_consistency(C.byref(p), fwd.contiguous().data_ptr(), bwd.contiguous().data_ptr(),
                                               float(tresh), out.data_ptr(), stream), "dflow_fb_consistency")
    return out
