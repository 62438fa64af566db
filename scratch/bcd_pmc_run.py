"""Small driver for rocprofv3 --pmc: B passes prepared, two batched sweeps."""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(0, 0))
a, b = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
dfs = []
for i in range(B):
    df = pl.DiscreteFlow(H, W, seed=i)
    df.load_pair(a, b); df.generisi(); df.nasumicni(); df.pakovanje()
    dfs.append(df)
torch.cuda.synchronize()
pl.ceoBCD_batch(dfs, 2)
torch.cuda.synchronize()
