"""CPU experiment (numpy + the oracle's DAISY): how many events would the kNN screen emit per (query, candidate cell) if the
matrix product ran over the first k principal directions only and the dropped directions were bounded by Cauchy-Schwarz
per group (one extra K slot per group: |q_g||c_g|)?  Exact arithmetic (f64), i.e. on top of the f16 rounding slack."""
import sys, os, importlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
H, W, ch, cw = 436, 1024, 27, 64
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cache = "/tmp/exp/desc_%d.npz" % seed
if os.path.exists(cache):
    z = np.load(cache); d1, d2 = z["d1"], z["d2"]
else:
    img1, img2, gt = synth.make_pair(H, W, seed=seed)
    d1, d2 = O.daisy(img1), O.daisy(img2)
    np.savez(cache, d1=d1, d2=d2)
d1 = d1.astype(np.float64); d2 = d2.astype(np.float64)
ncx, ncy = W // cw, H // ch
def x0(c): return c * cw
def x1(c): return W if c == ncx - 1 else (c + 1) * cw
def y0(c): return c * ch
def y1(c): return H if c == ncy - 1 else (c + 1) * ch
# PCA of image-2 descriptors (a sample)
rng = np.random.default_rng(1)
samp = d2.reshape(-1, 68)[rng.choice(H * W, 8192, replace=False)]
mu = samp.mean(0)
cov = np.cov((samp - mu).T)
ev, V = np.linalg.eigh(cov); order = np.argsort(-ev); ev = ev[order]; V = V[:, order]
print("spectrum: cumulative energy fraction left after k dims")
tot = ev.sum()
for k in (8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 60, 64):
    print("  k=%d: %.3e" % (k, ev[k:].sum() / tot))
r1 = (d1.reshape(-1, 68) - mu) @ V
r2 = (d2.reshape(-1, 68) - mu) @ V
r1 = r1.reshape(H, W, 68); r2 = r2.reshape(H, W, 68)
nq = 150
qs = [(int(rng.integers(0, H)), int(rng.integers(0, W))) for _ in range(nq)]
configs = [(68, []), (61, [7]), (60, [4, 4]), (45, [23]), (44, [12, 12]), (42, [8, 8, 10]), (29, [39]), (28, [20, 20]),
           (26, [14, 14, 14]), (24, [8, 8, 12, 16]), (22, [6, 8, 8, 10, 14]), (13, [55]), (12, [28, 28]), (10, [10, 16, 32]), (8, [6, 8, 12, 16, 18])]
res = {c[0]: [] for c in configs}
for (qy, qx) in qs:
    qc_i, qc_j = min(qx // cw, ncx - 1), min(qy // ch, ncy - 1)
    q = r1[qy, qx]
    for ci in range(max(0, qc_i - 2), min(ncx - 1, qc_i + 2) + 1):
        for cj in range(max(0, qc_j - 2), min(ncy - 1, qc_j + 2) + 1):
            c = r2[y0(cj):y1(cj), x0(ci):x1(ci)].reshape(-1, 68)
            diff2 = (c - q) ** 2
            d2full = diff2.sum(1)
            for (k, groups) in configs:
                dP = diff2[:, :k].sum(1)
                ub = dP.copy(); lb = dP.copy()
                s = k
                for gsz in groups:
                    nq_ = np.sqrt((q[s:s + gsz] ** 2).sum()); nc_ = np.sqrt((c[:, s:s + gsz] ** 2).sum(1))
                    ub += (nq_ + nc_) ** 2; lb += (nq_ - nc_) ** 2
                    s += gsz
                assert s == 68
                thr = np.partition(ub, 4)[4]
                res[k].append(int((lb <= thr).sum()))
for (k, groups) in configs:
    a = np.array(res[k])
    print("k=%2d groups=%-22s slots=%2d  events/(q,cell): mean %.2f p50 %d p90 %d p99 %d max %d" % (k, groups, k + len(groups), a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))
