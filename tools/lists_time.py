import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", sys.argv[1])
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
img1, img2, gt = synth.make_pair(H, W, seed=0)
df = pl.DiscreteFlow(H, W, seed=0)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni()
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
print(sys.argv[1:], "pakovanje %.3f ms" % t(df.pakovanje))
