"""BCD sweeps of P independent pairs at once on P streams (nothing else running): how far is the chain kernel from a
throughput regime?  Prints ms per (pair, sweep) for P = 1..6."""
import sys, os, importlib, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
PMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 6
img1, img2, gt = synth.make_pair(H, W, seed=2022)
a, b = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
dfs = []
for i in range(PMAX):
    df = pl.DiscreteFlow(H, W, seed=99 + i)
    df.load_pair(a, b); df.generisi(); df.nasumicni(); df.pakovanje()
    dfs.append(df)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(PMAX)]
for P in range(1, PMAX + 1):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                dfs[i].ceoBCD(2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("P=%d  wall %.3f ms for 2 sweeps  ->  %.3f ms per (pair, sweep)" % (P, dt, dt / (2 * P)), flush=True)
