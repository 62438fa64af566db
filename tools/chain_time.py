"""Batched BCD sweeps (8 passes per launch) and one pass alone: python tools/chain_time.py [variant.so]"""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", sys.argv[1])
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
dfs = []
for i in range(8):
    img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(i % 2, 0))
    df = pl.DiscreteFlow(H, W, seed=i)
    df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni(); df.pakovanje()
    dfs.append(df)
torch.cuda.synchronize()
def ev_time(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
t8 = ev_time(lambda: pl.ceoBCD_batch(dfs, 2)) / 16
t1 = ev_time(lambda: dfs[0].ceoBCD(2)) / 2
print(sys.argv[1:], "bcd ms per (pair, sweep): batched x8 %.3f, alone %.3f" % (t8, t1), "labels", int(dfs[0].bestlabels.sum().item()))
