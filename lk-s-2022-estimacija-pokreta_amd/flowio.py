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


# ------------------------------------------------------------------------------------------------ KITTI flow PNGs
# visualization.py:37-53 reads the KITTI ground truth with cv2.imread(path, -1): 16-bit RGB PNG, U = (R-32768)/64,
# V = (G-32768)/64, valid = B > 0.  cv2 is not a dependency of this build (and PIL cannot read 16-bit RGB), so the PNG
# container is decoded here: zlib + the five PNG scanline filters, non-interlaced truecolour, 8 or 16 bit.
def read_png16(path):
    """Returns (H,W,3) uint16 in R,G,B order (8-bit files are returned as their 8-bit values)."""
    import struct
    import zlib
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError("%s: no IHDR chunk" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 2 or depth not in (8, 16) or interlace != 0:
        raise ValueError("%s: only non-interlaced 8/16-bit RGB PNGs are supported (colour type %d, depth %d)" % (path, ctype, depth))
    bpp = 3 * depth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * bpp
    if raw.size != h * (stride + 1):
        raise ValueError("%s: truncated image data" % path)
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:                                     # Up
            cur = (line + prev) & 255
        elif ft == 1:                                     # Sub: running sum per byte lane of the pixel
            cur = (np.cumsum(line.reshape(w, bpp), axis=0) & 255).reshape(-1)
        elif ft in (3, 4):                                # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            lp, pv = line.tolist(), prev.tolist()
            res = [0] * stride
            for i in range(stride):
                a = res[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = pv[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                res[i] = (lp[i] + pred) & 255
            cur = np.array(res, np.int32)
        else:
            raise ValueError("%s: bad filter type %d" % (path, ft))
        out[y] = cur
        prev = cur
    if depth == 16:
        px = out.reshape(h, w, 3, 2).astype(np.uint16)
        return (px[..., 0] << 8) | px[..., 1]
    return out.reshape(h, w, 3).astype(np.uint16)


def write_png16(path, rgb16):
    """(H,W,3) uint16 R,G,B -> 16-bit truecolour PNG (filter type 0 on every line)."""
    import struct
    import zlib
    rgb16 = np.ascontiguousarray(rgb16, dtype=np.uint16)
    h, w, _ = rgb16.shape
    be = rgb16.astype(">u2").tobytes()
    stride = w * 6
    raw = b"".join(b"\x00" + be[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_kitti_flow_png(path):
    """FlowImage.readFlowFieldFromImage, visualization.py:37-53: (H,W,3) float32 [U,V,valid]."""
    img = read_png16(path).astype(np.float64)
    out = np.zeros(img.shape, np.float32)
    valid = img[..., 2] > 0
    out[..., 0] = np.where(valid, (img[..., 0] - 32768.0) / 64.0, 0.0)
    out[..., 1] = np.where(valid, (img[..., 1] - 32768.0) / 64.0, 0.0)
    out[..., 2] = valid
    return out


def write_kitti_flow_png(path, uvv):
    """(H,W,3) [U,V,valid] -> KITTI devkit layout (R = 64 U + 32768, G = 64 V + 32768, B = valid), the layout
    readFlowFieldFromImage parses.  (The reference's own writeFlowField, visualization.py:75-82, hands its U,V,1 array to
    cv2.imwrite as BGR, so its files have U in the blue channel and cannot be read back by its reader; not reproduced.)"""
    uvv = np.asarray(uvv)
    valid = uvv[..., 2] > 0.5
    img = np.zeros(uvv.shape, np.uint16)
    img[..., 0] = np.where(valid, uvv[..., 0].astype(np.float64) * 64.0 + 32768, 0).astype(np.uint16)
    img[..., 1] = np.where(valid, uvv[..., 1].astype(np.float64) * 64.0 + 32768, 0).astype(np.uint16)
    img[..., 2] = valid
    write_png16(path, img)
