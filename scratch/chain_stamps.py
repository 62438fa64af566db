"""Diagnostic builds of bcd_chain_kernel with s_memtime stamps (BCD_STAMP=k): mean cycles from a step's start to stamp k, per wave,
for one pass alone and for 7 passes per launch: python scratch/chain_stamps.py stampK.so"""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
_lib.LIB_PATH = os.path.join(ROOT, "scratch", "prof_build", sys.argv[1])
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
dfs[0].ceoBCD(1); torch.cuda.synchronize()
a = dfs[0].bestlabels.flatten()[:6].tolist()
pl.ceoBCD_batch(dfs, 1); torch.cuda.synchronize()
b = dfs[0].bestlabels.flatten()[:6].tolist()
print(sys.argv[1], "alone (row phase): cycles per wave", a[0], a[2], a[4], "steps", a[1], "| 7 passes:", b[0], b[2], b[4])
