import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H,W = 436,1024
img1,img2,gt = synth.make_pair(H,W,seed=2022)
df = pl.DiscreteFlow(H,W,seed=99)
df.load_pair(img1,img2); df.generisi(); torch.cuda.synchronize()
N=H*W
def al(x): return (x+255)//256*256
rows2 = 15*16*1728 + 16*2016   # image-2 rows in tile-position order, cells padded to 96 (knn_mfma.hip: km_total_rows)
off = (N+rows2)*160 + N*8
base = df.ws.data_ptr()
off = al(base+off)-base
ctr = df.ws[off:off+8].cpu().numpy().view(np.int32); print('ovf count, flags', ctr)
off += 256 + 512 + 8192*16
off = al(base+off)-base
ncx,ncy=16,16; qwaves=(64*31+63)//64; nl = ncx*ncy*qwaves*25
ev = df.ws[off:off+nl*4096*4]
cnt = df.ws[off+nl*4096*4: off+nl*4096*4+nl*128].cpu().numpy().reshape(nl,2,64)
# active lists: those with qwave*64 < qnpts and valid slot: approximate by cnt>0 any
act = cnt.reshape(nl,-1).max(1)>0
c = cnt[act].astype(np.int32)
print('lists', nl, 'active', act.sum(), 'entries/lane mean %.2f p99 %d max %d'%(c.mean(), np.percentile(c,99), c.max()))
evw = ev.cpu().numpy().view(np.uint32).reshape(nl,2,32,64)[act]
# count events = popcount of masks for valid entries
masks = evw & 0xFFFF
valid = np.arange(32)[None,None,:,None] < c[:,:,None,:]
pc = np.zeros(masks.shape, np.int32)
m = masks.copy()
for i in range(16): pc += (m>>i)&1
evs = (pc*valid).sum(2)   # per list, group, lane
print('events/lane mean %.2f p99 %d max %d ; events per query (2 lanes) mean %.2f'%(evs.mean(), np.percentile(evs,99), evs.max(), (evs[:,:,:32]+evs[:,:,32:]).mean()))
# resolve kernel (round 2): lane = query, rounds of a (wave, cell) = the largest event count among its 64 queries
q = (evs[:, :, :32] + evs[:, :, 32:]).reshape(len(evs), 64)
rounds = q.max(1)
print('per query events mean %.2f; rounds per (wave, cell) mean %.2f p50 %d p90 %d p99 %d max %d; lane utilisation %.1f %%' % (
    q.mean(), rounds.mean(), np.percentile(rounds, 50), np.percentile(rounds, 90), np.percentile(rounds, 99), rounds.max(), 100 * q.mean() / rounds.mean()))
h = np.bincount(np.minimum(q.reshape(-1), 40))
print('events per query histogram (0..40+):', h.tolist())
# if every lane walked its 5 cells' candidates back to back (no per-cell synchronisation): rounds per wave = max over lanes of the sum
# what pooling the tails (events beyond the 5th) of the cells of a window column would save: interior query cells only
allq = np.zeros((nl, 64), np.int32); allq[act] = q
lists = allq.reshape(ncx * ncy, qwaves, 25, 64)
cells = np.arange(ncx * ncy); ci, cj = cells % ncx, cells // ncx
inner = (ci >= 2) & (ci < ncx - 2) & (cj >= 2) & (cj < ncy - 2)
L = lists[inner][:, :qwaves - 1]                       # full waves only; [cell, qwave, slot = col*5 + row, query]
L = L.reshape(L.shape[0], L.shape[1], 5, 5, 64)        # [cell, qwave, col, row, query]
L = L[L.reshape(L.shape[0], L.shape[1], -1).max(-1) > 0]      # waves that hold queries: [wave, col, row, query]
t = np.maximum(L - 5, 0)
cur = L.max(-1).sum(-1)                                # rounds of a (wave, column) today
pair = 25 + (t[..., 0, :] + t[..., 1, :]).max(-1) + (t[..., 2, :] + t[..., 3, :]).max(-1) + t[..., 4, :].max(-1)
pool = 25 + t.sum(-2).max(-1)
print('rounds per (wave, window column): today %.2f, tails pooled in pairs of cells %.2f, tails of all 5 cells pooled %.2f, minimum %.2f'
      % (cur.mean(), pair.mean(), pool.mean(), L.sum(-2).mean()))
