"""Event statistics of the kNN screen (K = 48 layout of knn_mfma.hip): python scratch/ev_stats2.py [pair seed]"""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
KM_K = 48
img1, img2, gt = synth.make_pair(H, W, seed=seed)
df = pl.DiscreteFlow(H, W, seed=99)
df.load_pair(img1, img2); df.generisi(); torch.cuda.synchronize()
N = H * W
def al(x): return (x + 255) // 256 * 256
rows2 = 15 * 16 * 1728 + 16 * 2112          # cells padded to whole chunks of 192 rows
base = df.ws.data_ptr()
off = (N + rows2) * KM_K * 2 + N * 8
off = al(base + off) - base
ctr = df.ws[off:off + 8].cpu().numpy().view(np.int32); print('ovf count, flags', ctr)
off += 256 + 512 + 68 * 68 * 8
off = al(base + off) - base
off += 64 * 68 * 68 * 8
novf = int(ctr[0])
ovf = df.ws[off:off + min(novf, 8192) * 16].cpu().numpy().view(np.int32).reshape(-1, 4)
off += 8192 * 16
off = al(base + off) - base
ncx, ncy = 16, 16; qwaves = (64 * 31 + 63) // 64; nl = ncx * ncy * qwaves * 25
ev = df.ws[off:off + nl * 4096 * 4]
cnt = df.ws[off + nl * 4096 * 4: off + nl * 4096 * 4 + nl * 128].cpu().numpy().reshape(nl, 2, 64)
act = cnt.reshape(nl, -1).max(1) > 0
c = cnt[act].astype(np.int32)
print('lists', nl, 'active', act.sum(), 'overflowed lists', (c.reshape(len(c), -1).max(1) == 255).sum())
cc = np.where(c == 255, 32, c)
print('entries/lane mean %.2f p99 %d max %d' % (cc.mean(), np.percentile(cc, 99), cc.max()))
evw = ev.cpu().numpy().view(np.uint32).reshape(nl, 2, 32, 64)[act]
masks = evw & 0xFFFF
valid = np.arange(32)[None, None, :, None] < cc[:, :, None, :]
pc = np.zeros(masks.shape, np.int32)
for i in range(16): pc += (masks >> i) & 1
evs = (pc * valid).sum(2)
q = (evs[:, :, :32] + evs[:, :, 32:]).reshape(len(evs), 64)
print('events per (query, cell): mean %.2f p50 %d p90 %d p99 %d p99.9 %d max %d' % (q.mean(), *[np.percentile(q, x) for x in (50, 90, 99, 99.9)], q.max()))
rounds = q.max(1)
print('rounds per (wave, cell) mean %.2f; lane utilisation %.1f %%' % (rounds.mean(), 100 * q.mean() / rounds.mean()))
print('histogram 0..40+:', np.bincount(np.minimum(q.reshape(-1), 40)).tolist())
# union of the candidate rows the 64 queries of a wave ask for in one cell (resolve kernel: rows that could be staged once)
sel = np.flatnonzero(act)[::37][:3000]
evsel = ev.cpu().numpy().view(np.uint32).reshape(nl, 2, 32, 64)[sel]
csel = np.where(cnt[sel] == 255, 32, cnt[sel]).astype(np.int32)
un = []; tot = []
for li in range(len(sel)):
    ids = set(); n = 0
    for gq in range(2):
        for lane in range(64):
            for e in range(csel[li, gq, lane]):
                w = int(evsel[li, gq, e, lane]); tile = w >> 16; m = w & 0xFFFF; h = lane >> 5
                while m:
                    r = (m & -m).bit_length() - 1; m &= m - 1
                    ids.add(((4 * h + (r & 3) + 8 * (r >> 2)), tile)); n += 1
    un.append(len(ids)); tot.append(n)
un = np.array(un); tot = np.array(tot)
print('per (wave, cell): events %.1f, distinct candidate rows %.1f (p50 %d p90 %d p99 %d max %d); lists with <= 64 distinct rows: %.1f %%, <= 128: %.1f %%' % (
    tot.mean(), un.mean(), *[np.percentile(un, x) for x in (50, 90, 99)], un.max(), 100 * (un <= 64).mean(), 100 * (un <= 128).mean()))
if novf:
    qc = ovf[:, 0]; print('overflow entries by query cell row (cj):', np.bincount(qc // ncx, minlength=ncy).tolist())
    print('by query cell column (ci):', np.bincount(qc % ncx, minlength=ncx).tolist())
    print('by candidate cell row:', np.bincount(ovf[:, 3], minlength=ncy).tolist(), 'column:', np.bincount(ovf[:, 2], minlength=ncx).tolist())
