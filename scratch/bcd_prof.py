"""Diagnostic: per-section cycle breakdown of bcd_chain_kernel's step (needs scratch/prof_build/libdflow_prof.so, built with -DBCD_PROF)."""
import sys, os, importlib, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
_lib.LIB_PATH = os.path.join(ROOT, "scratch", "prof_build", "libdflow_prof.so")
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
img1, img2, gt = synth.make_pair(H, W, seed=2022)
df = pl.DiscreteFlow(H, W, seed=99)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni(); df.pakovanje()
L = _lib.lib()
names = ["issue reads", "precompute+fetch", "first 8", "next 4+4", "residual", "dp write", "wave min", "barrier"]
for ph in (1, 0):
    torch.cuda.synchronize(); L.dflow_debug_bcd_prof_reset()
    df.bcd_phase(ph); torch.cuda.synchronize()
    out = (C.c_ulonglong * 24)(); L.dflow_debug_bcd_prof(out)
    a = np.array(out[:]).reshape(3, 8).astype(float)
    steps = (W if ph in (1, 3) else H) - 1
    print("phase", ph, "steps", steps)
    for k, n in enumerate(names):
        print("  %-12s" % n, " ".join("%8.1f" % (a[wv, k] / steps) for wv in range(3)))
    print("  %-12s" % "total", " ".join("%8.1f" % (a[wv].sum() / steps) for wv in range(3)))
