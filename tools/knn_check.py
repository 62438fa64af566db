"""MFMA-screened kNN of a variant library against its brute-force kernel, every pixel: python tools/knn_check.py [variant.so] [H W cellh cellw]"""
import sys, os, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
args = sys.argv[1:]
if args and args[0].endswith(".so"): _lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", args.pop(0))
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
geoms = [(436, 1024, 27, 64), (375, 1242, 25, 54), (97, 131, 9, 13)] if not args else [tuple(int(v) for v in args[:4])]
for H, W, ch, cw in geoms:
    df = pl.DiscreteFlow(H, W, cellh=ch, cellw=cw, seed=5)
    img1, img2, _ = synth.make_pair(H, W, seed=77)
    df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda())
    out = {}
    for mode in (0, _lib.FLAG_KNN_EXACT):
        df.p.flags = mode
        df.generisi(); torch.cuda.synchronize()
        out[mode] = [t.clone() for t in (df.proposals, df.lcosts, df.nprop, df.bestlabels)]
    ok = all(torch.equal(a, b) for a, b in zip(out[0], out[_lib.FLAG_KNN_EXACT]))
    print((H, W, ch, cw), "identical" if ok else "DIFFERENT")
    if not ok: sys.exit(1)
