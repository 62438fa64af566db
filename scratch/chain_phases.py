"""One batched sweep (7 passes per launch) and one pass alone, for a rocprofv3 --kernel-trace run: per-phase durations."""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
dfs = []
for i in range(7):
    img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(i % 2, 0))
    df = pl.DiscreteFlow(H, W, seed=i)
    df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni(); df.pakovanje()
    dfs.append(df)
torch.cuda.synchronize()
for b in range(1, 8):
    pl.ceoBCD_batch(dfs[:b], 1); torch.cuda.synchronize()
