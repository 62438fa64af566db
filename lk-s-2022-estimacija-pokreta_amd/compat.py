"""The reference's compat-matrix files for users who feed its own BCD scripts:
packedksets (pakovanje, daisy i flann.py:256-309) and the four scan-order copies of pakovanjeZaC (:321-398).
The bit matrices come from the GPU (dflow_pack_compat); the host part replays the reference's scratch reuse on the
H+W border matrices and re-indexes.  Not on the hot path: the GPU BCD never materialises these 2.6 GB."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def packedksets(df):
    """df: a DiscreteFlow whose proposals are final.  Returns (H,W,2,kdim) uint8 exactly as
    'Daisy output slike ... packedksets.npy' (daisy i flann.py:308)."""
    H, W, L = df.p.pich, df.p.picw, df.p.maxnprop
    kdim = L * L // 8 + 1
    dev = torch.empty((H, W, 2, kdim), dtype=torch.uint8, device=df.device)
    stream = C.c_void_p(torch.cuda.current_stream(df.device).cuda_stream)
    _lib.check(_lib.lib().dflow_pack_compat(C.byref(df.p), df.proposals.data_ptr(), df.nprop.data_ptr(), dev.data_ptr(), stream),
               "dflow_pack_compat")
    packed = dev.cpu().numpy()
    replay_border_scratch(packed, df.nprop.cpu().numpy(), L)
    return packed


def replay_border_scratch(packed, nprop, L):
    """The loops over the bottom row (daisy i flann.py:283-293) and the right column (:294-307) never clear ksets4:
    entries outside [0:nprop[pixel], 0:nprop[neighbour]] keep what earlier pixels of the same loop left there."""
    H, W = nprop.shape
    for slot, pixels in ((1, [(H - 1, tx, H - 1, tx + 1) for tx in range(W - 1)]),
                         (0, [(ty, W - 1, ty + 1, W - 1) for ty in range(H - 1)])):
        scratch = np.zeros((L, L), bool)
        for (ty, tx, ny, nx) in pixels:
            fresh = np.unpackbits(packed[ty, tx, slot])[:L * L].reshape(L, L).astype(bool)
            r, c = int(nprop[ty, tx]), int(nprop[ny, nx])
            scratch[:r, :c] = fresh[:r, :c]
            packed[ty, tx, slot] = np.packbits(scratch.reshape(-1))


def pakovani_za_c(packed):
    """pakovanjeZaC, daisy i flann.py:321-398: the same matrices indexed [chain, step] in the scan order of the four
    BCD phases.  Returns the four arrays of 'Daisy output slike ... pakovani za c {0..3}.npy'."""
    H, W, _, kdim = packed.shape
    p0 = np.zeros(((W + 1) // 2, H, kdim), np.uint8); p2 = np.zeros_like(p0)
    p1 = np.zeros(((H + 1) // 2, W, kdim), np.uint8); p3 = np.zeros_like(p1)
    ty = np.arange(H - 1)
    for tx in range(W):                                   # slot 0 exists for ty < H-1
        if tx % 2 == 0:
            p0[tx // 2, ty] = packed[ty, tx, 0]
        else:
            p2[(W - 1 - tx) // 2, H - 2 - ty] = packed[ty, tx, 0]
    tx = np.arange(W - 1)
    for y in range(H):                                    # slot 1 exists for tx < W-1
        if y % 2 == 0:
            p1[y // 2, W - 2 - tx] = packed[y, tx, 1]
        else:
            p3[(H - 1 - y) // 2, tx] = packed[y, tx, 1]
    return p0, p1, p2, p3


def remove_small_segments(sparse, tresh, min_segment_size):
    """removeSmallSegments, postprocessing.py:29-76, in place on a host (A,B,3) float32 [U,V,valid] field."""
    if not (isinstance(sparse, np.ndarray) and sparse.dtype == np.float32 and sparse.ndim == 3 and sparse.shape[2] == 3
            and sparse.flags.c_contiguous):
        raise ValueError("sparse must be a C-contiguous float32 (A,B,3) array")
    _lib.check(_lib.lib().dflow_remove_small_segments_host(sparse.ctypes.data, sparse.shape[0], sparse.shape[1],
                                                           float(tresh), int(min_segment_size)),
               "dflow_remove_small_segments_host")
    return sparse
