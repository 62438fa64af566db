"""On-disk contract of the hot path: the reference's .npy names/dtypes/[dy,dx] order,
plus Middlebury .flo ([u=dx, v=dy] float32) as parsed by visualization.py:9-29.
"""
import numpy as np

FLO_MAGIC = np.float32(202021.25)


def pic_prefix(picindex):
    """'1' + two-digit pair index (daisy i flann.py:16,24-25,201)."""
    s = str(picindex)
    if len(s) == 1:
        s = "0" + s
    return "1" + s


def flow_name(picindex, backward, sweep):
    """daisy i flann.py:201 / python bcd.py:282."""
    return "Gotova flow slika %s backward=%s posle %02d BCD.npy" % (pic_prefix(picindex), backward, sweep)


def labels_name(picindex, backward, sweep):
    """daisy i flann.py:202 / python bcd.py:283."""
    return "Bestlabels fajl slike %s backward=%s posle %02d BCD.npy" % (pic_prefix(picindex), backward, sweep)


def stage_name(picindex, backward, what):
    """daisy i flann.py:251-253,308: what in {proposals_nakon_gausa, lcosts_nakon_gausa, nprop, packedksets}."""
    return "Daisy output slike %s backward=%s %s.npy" % (pic_prefix(picindex), backward, what)


def write_flo(path, flow_dydx):
    """(H,W,2) [dy,dx] -> Middlebury .flo: f32 magic, i32 w, i32 h, then h*w*2 f32 [u,v] row-major."""
    flow_dydx = np.asarray(flow_dydx)
    h, w, _ = flow_dydx.shape
    uv = np.ascontiguousarray(flow_dydx[..., ::-1], dtype=np.float32)
    with open(path, "wb") as f:
        FLO_MAGIC.tofile(f)
        np.array([w, h], np.int32).tofile(f)
        uv.tofile(f)


def read_flo(path):
    """Returns (H,W,2) float32 [u,v]."""
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, 1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise ValueError("%s: bad .flo magic" % path)
        w, h = (int(v) for v in np.fromfile(f, np.int32, 2))
        data = np.fromfile(f, np.float32, 2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError("%s: truncated .flo" % path)
    return data.reshape(h, w, 2)
