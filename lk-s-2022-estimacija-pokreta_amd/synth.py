"""Synthetic image pairs with known ground-truth flow: the reference reads KITTI PNGs from
../data_scene_flow (daisy i flann.py:26-27), which do not exist here, so tests and bench.py use these.

img1: multi-octave smoothed noise texture, uint8 BGR.  gt: smooth flow, sum of low-frequency sinusoids,
|u| <= amp_x, |v| <= amp_y (inside the +-2-cell search window).  img2(p + gt(p)) ~= img1(p): img2 is img1
forward-mapped by the flow, realised as a backward warp with the (smooth) flow sampled at the target.
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


def gt_flow(H, W, seed, amp_x=40.0, amp_y=20.0):
    """(H,W,2) float64 [dy,dx] ground truth."""
    rng = np.random.default_rng(seed + 7919)
    yy, xx = np.meshgrid(np.arange(H) / H, np.arange(W) / W, indexing="ij")
    u = np.zeros((H, W)); v = np.zeros((H, W))
    for k in range(3):
        fx, fy = rng.uniform(0.5, 2.0, 2)
        ph = rng.uniform(0, 2 * np.pi, 2)
        u += np.sin(2 * np.pi * (fx * xx + fy * yy) + ph[0]) / 3.0
        v += np.cos(2 * np.pi * (fy * xx - fx * yy) + ph[1]) / 3.0
    return np.stack([amp_y * v, amp_x * u], axis=-1)


def make_pair(H, W, seed=0, amp_x=40.0, amp_y=20.0, noise=2.0):
    """Returns (img1, img2, gt) with img uint8 (H,W,3) BGR and gt (H,W,2) float64 [dy,dx]."""
    rng = np.random.default_rng(seed)
    chans = [_smooth_noise(rng, H, W) for _ in range(3)]
    img1 = np.stack(chans, axis=-1) * 255.0
    gt = gt_flow(H, W, seed, amp_x, amp_y)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    sy = np.clip(yy - gt[..., 0], 0, H - 1.001); sx = np.clip(xx - gt[..., 1], 0, W - 1.001)
    y0 = np.floor(sy).astype(int); x0 = np.floor(sx).astype(int)
    fy = (sy - y0)[..., None]; fx = (sx - x0)[..., None]
    img2 = ((1 - fy) * ((1 - fx) * img1[y0, x0] + fx * img1[y0, x0 + 1])
            + fy * ((1 - fx) * img1[y0 + 1, x0] + fx * img1[y0 + 1, x0 + 1]))
    img2 = img2 + rng.normal(0.0, noise, img2.shape)
    to_u8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to_u8(img1), to_u8(img2), gt


def pair_seed(pair_idx, backward):
    """Seed convention of SURVEY 8(d): 1000 * pair_idx + direction."""
    return 1000 * int(pair_idx) + int(backward)
