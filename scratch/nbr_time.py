"""neighbour_kernel alone on one pair: python scratch/nbr_time.py [variant.so]"""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "scratch", "prof_build", sys.argv[1])
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
img1, img2, gt = synth.make_pair(H, W, seed=2022)
df = pl.DiscreteFlow(H, W, seed=99)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda())
ts = []
for _ in range(4):
    df.generisi(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); df.nasumicni(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(sys.argv[1:], "neighbour %.3f ms" % min(ts[1:]), "nprop sum", int(df.nprop.sum().item()) if hasattr(df, "nprop") else "")
