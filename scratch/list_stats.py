"""Compat-list length statistics at full size from the proposals themselves (torch on the GPU, a band of image rows):
per-label counts for both chain directions, and what a 64-label wave sees (max over its lanes)."""
import sys, os, importlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W, LP = 436, 1024, 160
img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(0, 0))
df = pl.DiscreteFlow(H, W, seed=0)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni()
torch.cuda.synchronize()
p = df.proposals.view(torch.int32)
dy = ((p << 16) >> 16).to(torch.int16); dx = (p >> 16).to(torch.int16)
npr = df.nprop
lab = torch.arange(LP, device="cuda")
for name, (oy, ox) in (("dir0 (vertical)", (1, 0)), ("dir1 (horizontal)", (0, 1))):
    cnts = []
    for y0 in range(100, 132, 4):
        a = slice(y0, y0 + 4); b = slice(y0 + oy, y0 + 4 + oy)
        xa = slice(0, W - ox); xb = slice(ox, W)
        d = (dy[a, xa, :, None] - dy[b, xb, None, :]).abs() + (dx[a, xa, :, None] - dx[b, xb, None, :]).abs()
        ok = (d < 8) & (lab[None, None, :, None] < npr[a, xa, None, None]) & (lab[None, None, None, :] < npr[b, xb, None, None])
        cnts.append(ok.sum(-1))
    c = torch.cat(cnts, 0)                                   # (rows, W', LP)
    valid = lab[None, None, :] < torch.cat([npr[y0:y0 + 4, 0:W - ox] for y0 in range(100, 132, 4)], 0)[..., None]
    cv = c[valid].cpu().numpy()
    print(name, "labels", cv.size, "mean %.2f" % cv.mean(), " ".join(">%d: %.3f" % (t, (cv > t).mean()) for t in (0, 4, 5, 6, 8, 10, 12, 15, 16, 20)))
    print("  hist", np.bincount(np.minimum(cv, 24), minlength=25))
    c = torch.where(valid, c, torch.zeros_like(c))
    for wv in range(3):
        mx = c[..., 64 * wv:64 * wv + 64].amax(-1).cpu().numpy().ravel()
        print("  wave %d: max-count mean %.2f " % (wv, mx.mean()) + " ".join("any>%d: %.3f" % (t, (mx > t).mean()) for t in (5, 8, 10, 12, 15, 16, 20)))
    mx = c.amax(-1).cpu().numpy().ravel()
    print("  block: " + " ".join("any>%d: %.3f" % (t, (mx > t).mean()) for t in (5, 8, 10, 12, 15, 16, 20)))
    # owners of a second (> 5 members) / third (> 10) block per 64-label wave: how often do BCD_CAP_B / BCD_CAP_C slots not suffice?
    for wv in range(3):
        nb = (c[..., 64 * wv:64 * wv + 64] > 5).sum(-1).cpu().numpy().ravel(); nc = (c[..., 64 * wv:64 * wv + 64] > 10).sum(-1).cpu().numpy().ravel()
        nm = (c[..., 64 * wv:64 * wv + 64] > 15).sum(-1).cpu().numpy().ravel()
        print("  wave %d owners: B mean %.1f " % (wv, nb.mean()) + " ".join(">%d: %.4f" % (t, (nb > t).mean()) for t in (24, 32, 40, 48, 56)) +
              " | C mean %.2f " % nc.mean() + " ".join(">%d: %.4f" % (t, (nc > t).mean()) for t in (4, 8, 12, 16, 24, 32)) + " | any>15: %.4f" % (nm > 0).mean())
