"""Synthetic image pairs with known ground-truth flow: the reference reads KITTI PNGs from
../data_scene_flow (daisy i flann.py:26-27), which do not exist here, so tests and bench.py use these.

img1: multi-octave smoothed noise texture, uint8 BGR.  g: smooth warp field, sum of low-frequency sinusoids,
|gx| <= amp_x, |gy| <= amp_y (inside the +-2-cell search window).  img2(q) = img1(q - g(q)) (bilinear backward warp)
+ noise; the ground-truth flow of image 1's pixels is the fixed point f(p) = g(p + f(p)) (forward_gt).
"""
import numpy as np


def _smooth_noise(rng, H, W, octaves=3):
    acc = np.zeros((H, W), np.float64)
    for o in range(octaves):
        step = 2 ** (octaves - o + 1)                     # 16, 8, 4 px blobs
        h, w = H // step + 3, W // step + 3
        coarse = rng.standard_normal((h, w))
        ys = np.arange(H) / step + 1.0
        xs = np.arange(W) / step + 1.0
        y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = coarse[y0][:, x0]; b = coarse[y0][:, x0 + 1]
        c = coarse[y0 + 1][:, x0]; d = coarse[y0 + 1][:, x0 + 1]
        acc += ((1 - fy) * ((1 - fx) * a + fx * b) + fy * ((1 - fx) * c + fx * d)) / (o + 1)
    acc -= acc.min()
    acc /= max(acc.max(), 1e-9)
    return acc


def _field(seed, amp_x, amp_y):
    """The analytic warp field g(y, x) -> (gy, gx) in pixels of pair `seed`, y and x in units of the image size."""
    rng = np.random.default_rng(seed + 7919)
    terms = []
    for k in range(3):
        fx, fy = rng.uniform(0.5, 2.0, 2)
        ph = rng.uniform(0, 2 * np.pi, 2)
        terms.append((fx, fy, ph[0], ph[1]))

    def g(yn, xn):
        u = np.zeros_like(xn, dtype=np.float64); v = np.zeros_like(xn, dtype=np.float64)
        for fx, fy, p0, p1 in terms:
            u = u + np.sin(2 * np.pi * (fx * xn + fy * yn) + p0) / 3.0
            v = v + np.cos(2 * np.pi * (fy * xn - fx * yn) + p1) / 3.0
        return amp_y * v, amp_x * u
    return g


def gt_flow(H, W, seed, amp_x=40.0, amp_y=20.0):
    """(H,W,2) float64 [dy,dx]: the warp field sampled on the pixel grid of IMAGE 2 (img2(q) = img1(q - g(q)))."""
    yy, xx = np.meshgrid(np.arange(H) / H, np.arange(W) / W, indexing="ij")
    gy, gx = _field(seed, amp_x, amp_y)(yy, xx)
    return np.stack([gy, gx], axis=-1)


def forward_gt(H, W, seed, amp_x=40.0, amp_y=20.0, iters=60):
    """(H,W,2) float64 [dy,dx]: the flow of IMAGE 1's pixels, i.e. what the hot path estimates.  img2 is a backward warp
    (img2(q) = img1(q - g(q))), so pixel p of image 1 reappears at q = p + f(p) with f(p) = g(q): the fixed point of
    f <- g(p + f), iterated on the analytic field (a contraction: |grad g| < 1 for the amplitudes used here).  Using g(p)
    itself as ground truth is off by |grad g| * |g|, several pixels at 40 px amplitude."""
    g = _field(seed, amp_x, amp_y)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    fy = np.zeros((H, W)); fx = np.zeros((H, W))
    for _ in range(iters):
        fy, fx = g((yy + fy) / H, (xx + fx) / W)
    return np.stack([fy, fx], axis=-1)


def _warp(img, gt):
    """Bilinear backward warp: out(q) = img(q - gt(q)), source clamped to the image."""
    H, W = gt.shape[:2]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    sy = np.clip(yy - gt[..., 0], 0, H - 1.001); sx = np.clip(xx - gt[..., 1], 0, W - 1.001)
    y0 = np.floor(sy).astype(int); x0 = np.floor(sx).astype(int)
    fy = (sy - y0)[..., None]; fx = (sx - x0)[..., None]
    return ((1 - fy) * ((1 - fx) * img[y0, x0] + fx * img[y0, x0 + 1])
            + fy * ((1 - fx) * img[y0 + 1, x0] + fx * img[y0 + 1, x0 + 1]))


def _box_blur(a, r, times=3):
    """`times` box filters of radius r along both axes (close to a Gaussian of sigma r * sqrt(times / 3)), edges replicated."""
    for _ in range(times):
        for ax in (0, 1):
            pad = [(0, 0)] * a.ndim
            pad[ax] = (r + 1, r)
            c = np.cumsum(np.pad(a, pad, mode="edge"), axis=ax)
            n = a.shape[ax]
            hi = np.take(c, np.arange(2 * r + 1, 2 * r + 1 + n), axis=ax)
            lo = np.take(c, np.arange(0, n), axis=ax)
            a = (hi - lo) / (2 * r + 1)
    return a


STYLES = ("dense", "low_texture")


def low_texture_regions(H, W, seed=0):
    """Region map of the "low_texture" style, (H,W) uint8: 0 dense texture, 1 saturated sky (exactly 255: exactly-zero DAISY
    descriptors, daisy i flann.py:66 uses NRM_NONE), 2 low-contrast road (a texture of +-2 grey levels, noise below one
    grey level), 3 blurred, 4 a pattern that repeats every 16 px.  What a KITTI frame has and smoothed noise has not."""
    yy, xx = np.meshgrid(np.arange(H) / H, np.arange(W) / W, indexing="ij")
    ph = 2 * np.pi * ((seed % 97) / 97.0)
    reg = np.zeros((H, W), np.uint8)
    reg[yy < 0.30 + 0.05 * np.sin(2 * np.pi * 1.5 * xx + ph)] = 1                 # sky: about 30 % of the frame
    reg[(yy > 0.62 + 0.03 * np.cos(2 * np.pi * xx + ph)) & (yy < 0.86)] = 2       # road band: about 24 %
    reg[(yy > 0.36) & (yy < 0.58) & (xx > 0.06) & (xx < 0.30)] = 3                 # blurred box
    reg[(yy > 0.36) & (yy < 0.58) & (xx > 0.64) & (xx < 0.90)] = 4                 # repeated pattern
    return reg


def make_pair(H, W, seed=0, amp_x=40.0, amp_y=20.0, noise=2.0, style="dense"):
    """Returns (img1, img2, gt) with img uint8 (H,W,3) BGR and gt (H,W,2) float64 [dy,dx] = forward_gt: the true flow of
    image 1's pixels (where the warp leaves the image the field is still defined; the clipped source has no match there).
    style "dense": texture everywhere (the best case for the kNN screen); "low_texture": low_texture_regions() on top."""
    if style not in STYLES:
        raise ValueError("style must be one of %s" % (STYLES,))
    rng = np.random.default_rng(seed)
    chans = [_smooth_noise(rng, H, W) for _ in range(3)]
    img1 = np.stack(chans, axis=-1) * 255.0
    gt = gt_flow(H, W, seed, amp_x, amp_y)
    sig = np.full((H, W, 1), float(noise))
    if style == "low_texture":
        reg = low_texture_regions(H, W, seed)[..., None]
        blurred = _box_blur(img1, 6)
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        pattern = (128.0 + 60.0 * np.sin(2 * np.pi * xx / 16.0) * np.cos(2 * np.pi * yy / 16.0))[..., None] * np.ones(3)
        img1 = np.where(reg == 1, 400.0, img1)                                 # far beyond 255: saturated after the warp and the noise too
        img1 = np.where(reg == 2, 90.0 + (img1 - 127.5) * (2.0 / 127.5), img1)
        img1 = np.where(reg == 3, blurred, img1)
        img1 = np.where(reg == 4, pattern, img1)
        sig = np.where(reg == 2, 0.4, sig)
    img2 = _warp(img1, gt) + rng.normal(0.0, 1.0, img1.shape) * _warp(sig, gt)
    to_u8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to_u8(img1), to_u8(img2), forward_gt(H, W, seed, amp_x, amp_y)


def pair_seed(pair_idx, backward):
    """Seed convention of SURVEY 8(d): 1000 * pair_idx + direction."""
    return 1000 * int(pair_idx) + int(backward)
