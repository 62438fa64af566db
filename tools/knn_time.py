"""kNN stage alone on the bench's two pairs and on seed 2022: python tools/knn_time.py [variant.so]"""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", sys.argv[1])
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
df = pl.DiscreteFlow(H, W, seed=99)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
out = []
for sd in (synth.pair_seed(0, 0), synth.pair_seed(1, 0), 2022):
    img1, img2, gt = synth.make_pair(H, W, seed=sd)
    df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda())
    out.append("%.3f" % t(df.generisi))
print(sys.argv[1:], "knn ms (bench pair 0, bench pair 1, seed 2022):", " ".join(out))
