"""List-length statistics of the BCD label records at full size (reads the workspace after pakovanje)."""
import sys, os, importlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W, LP = 436, 1024, 160
img1, img2, gt = synth.make_pair(H, W, seed=2022)
df = pl.DiscreteFlow(H, W, seed=99)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni(); df.pakovanje()
torch.cuda.synchronize()
def back_bytes():
    m = 0
    for n, ln in (((W + 1) // 2, H), ((H + 1) // 2, W), (W // 2, H), (H // 2, W)):
        m = max(m, n * ln * LP)
    return ((m + 255) & ~255) + 256
off = back_bytes() + H * W * 2 * LP * 5 * 4
rows = slice(100, 140)                       # a band of image rows
nprop = df.nprop.cpu().numpy()
recs = df.ws[off + rows.start * W * 2 * LP * 32: off + rows.stop * W * 2 * LP * 32].cpu().numpy().view(np.uint32).reshape(rows.stop - rows.start, W, 2, LP, 8)
lists = recs[..., :4].copy().view(np.uint8).reshape(rows.stop - rows.start, W, 2, LP, 16)
n = (lists != 0xFF).sum(-1)
more = (recs[..., 5] >> 31) != 0
valid = np.broadcast_to(np.arange(LP)[None, None, None, :] < nprop[rows][:, :, None, None], n.shape)
n = np.where(valid, n, 0); more = more & valid
print("rows", valid.sum(), "mean n", n[valid].mean(), "frac>4", (n[valid] > 4).mean(), "frac>8", (n[valid] > 8).mean(), "frac>12", (n[valid] > 12).mean(), "frac more", more[valid].mean())
print("hist n", np.bincount(n[valid], minlength=17))
for wv in range(3):
    mx = n[..., 64 * wv:64 * wv + 64].max(-1)
    mm = more[..., 64 * wv:64 * wv + 64].any(-1)
    print("wave", wv, "max-n hist", np.bincount(mx.ravel(), minlength=17), "any>8 %.3f any>12 %.3f any more %.3f" % ((mx > 8).mean(), (mx > 12).mean(), mm.mean()))
print("block any more %.3f" % more.any(-1).mean())
