"""Which stages overlap with a batched BCD (8 passes, 4 sweeps)?  Wall time of A alone, B alone, A || B on two streams."""
import sys, os, importlib, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
img1, img2, gt = synth.make_pair(H, W, seed=0)
a, b = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
bcd = []
for i in range(8):
    df = pl.DiscreteFlow(H, W, seed=i); df.load_pair(a, b); df.generisi(); df.nasumicni(); df.pakovanje(); bcd.append(df)
fe = []
for i in range(4):
    df = pl.DiscreteFlow(H, W, seed=i); df.load_pair(a, b); df.generisi(); df.nasumicni(); fe.append(df)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def wall(fa, fb, n=2):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
A = lambda: pl.ceoBCD_batch(bcd, 4)
def knn8():
    for k in range(8): fe[k % 4].generisi()
def lists8():
    for k in range(8): fe[k % 4].pakovanje()
def nbr8():
    for k in range(8):
        fe[k % 4].nasumicni()
tA = wall(A, None)
for name, B in (("8 x generisi", knn8), ("8 x pakovanje", lists8), ("8 x nasumicni", nbr8)):
    tB = wall(None, B); tAB = wall(A, B)
    print("BCD batch %.1f ms | %s %.1f ms | together %.1f ms (sum %.1f, max %.1f)" % (tA, name, tB, tAB, tA + tB, max(tA, tB)), flush=True)
