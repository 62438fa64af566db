"""Consumers on the far side of the hot path, host side: the EpicFlow match exporter of
napravi_parove.py and the error metrics of visualization.py.  Plain numpy on small (H,W,3) fields; not on the hot path.
"""
import numpy as np


def parovi(sparse_field, txtfile):
    """napravi_parove.py:3-13: one line "u v u+U v+V" per valid pixel of a (H,W,3) float32 [U,V,valid] field, in
    row-major order, numbers formatted by str() exactly as the reference does (python int for u, v; numpy float32 for
    the sums)."""
    flow = np.asarray(sparse_field)
    height, width, _ = flow.shape
    with open(txtfile, "w+") as f:
        for v in range(height):
            row = flow[v]
            for u in range(width):
                if row[u][2] > 0.5:
                    f.write(str(u) + ' ' + str(v) + ' ' + str(row[u][0] + u) + ' ' + str(row[u][1] + v) + '\n')


def to_uv_valid(flow, valid=None):
    """(H,W,2) [dy,dx] (the hot path's .npy layout, visualization.py:109-124) -> (H,W,3) float32 [U,V,valid]."""
    flow = np.asarray(flow)
    out = np.zeros(flow.shape[:2] + (3,), np.float32)
    out[..., 0] = flow[..., 1]
    out[..., 1] = flow[..., 0]
    out[..., 2] = 1.0 if valid is None else np.asarray(valid, np.float32)
    return out


def error_metrics(test_uvv, gt_uvv, abs_thresh=3.0):
    """errorImage, visualization.py:128-152, without the colour image: mean end-point error over the pixels valid in
    both fields and the percentage of them with EPE > 3 px.  Inputs are (H,W,3) float32 [U,V,valid] fields; the
    arithmetic is float32 like the reference's (np.sqrt of float32 products), the average is numpy's."""
    t = np.asarray(test_uvv, np.float32)
    g = np.asarray(gt_uvv, np.float32)
    both = (g[..., 2] > 0.5) & (t[..., 2] > 0.5)        # isValid, visualization.py:64-65
    dfu = t[..., 0][both] - g[..., 0][both]
    dfv = t[..., 1][both] - g[..., 1][both]
    err = np.sqrt(dfu * dfu + dfv * dfv)
    n = int(err.size)
    mean_epe = float(np.average(err)) if n else float("nan")
    outliers = float((err > abs_thresh).sum() * 100 / n) if n else float("nan")
    return mean_epe, outliers, n


def ucitajFlow(path):
    """FlowImage.ucitajFlow, visualization.py:97-124: a flow file -> (H,W,3) float32 [U,V,valid].  '.png' = KITTI ground
    truth (16-bit, :37-53), '.npy' = the hot path's [dy,dx] fields or a [U,V,valid]-like 3-channel field read with the same
    channel swap the reference applies (:104-110), '.flo' = Middlebury [u,v] (:118-124)."""
    import os
    from . import flowio
    ext = os.path.splitext(path)[1]
    if ext == ".png":
        return flowio.read_kitti_flow_png(path)
    if ext == ".npy":
        flow = np.load(path)
        out = np.zeros(flow.shape[:2] + (3,), np.float32)
        out[..., 0] = flow[..., 1]
        out[..., 1] = flow[..., 0]
        out[..., 2] = (flow[..., 2] != 0) if flow.shape[2] == 3 else 1.0      # setValid(val): "if val" (:55-57)
        return out
    if ext == ".flo":
        uv = flowio.read_flo(path)
        out = np.ones(uv.shape[:2] + (3,), np.float32)
        out[..., :2] = uv
        return out
    raise ValueError("unsupported flow file: %s" % path)
