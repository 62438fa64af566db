"""Random geometries (image size, cell size, window) through kNN -> neighbour -> one sweep, GPU vs oracle, bit for bit.
python scratch/fuzz_geoms.py [n] [seed] [f16]      (f16: binary16 descriptor planes, DFLOW_FLAG_DESCR_F16)"""
import sys, os, importlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F16 = len(sys.argv) > 3 and sys.argv[3] == "f16"
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
bad = 0
for it in range(n):
    ch, cw = int(rng.integers(3, 20)), int(rng.integers(3, 24))
    if ch * cw < 5: cw = 5
    ncy, ncx = int(rng.integers(1, 8)), int(rng.integers(1, 9))
    H, W = ch * ncy + int(rng.integers(0, ch)), cw * ncx + int(rng.integers(0, cw))
    H, W = max(H, 24), max(W, 24)
    window = int(rng.integers(0, 3))
    ngauss = int(rng.choice([0, 7, 25]))
    maxnprop = 5 * (2 * window + 1) ** 2 + ngauss
    over = dict(window=window, ngauss=ngauss, maxnprop=max(maxnprop, 5 + ngauss))
    try:
        df = pl.DiscreteFlow(H, W, ch, cw, seed=it, flags=_lib.FLAG_DESCR_F16 if F16 else 0, **over)
    except Exception as e:
        print("skip", (H, W, ch, cw, over), str(e)[:80]); continue
    p = O.make_params(H, W, ch, cw, seed=it, **over)
    img1, img2, _ = synth.make_pair(H, W, seed=100 + it, amp_x=0.08 * W, amp_y=0.08 * H)
    df.load_pair(img1, img2)
    d1, d2 = O.daisy(img1), O.daisy(img2)
    if F16:
        d1, d2 = d1.astype(np.float16).astype(np.float32), d2.astype(np.float16).astype(np.float32)
    df.generisi()
    pr, lc, npr, bl = O.knn_proposals(p, d1, d2)
    st = df.host_state()
    ok = all(np.array_equal(st[k], v) for k, v in (("nprop", npr), ("proposals", pr), ("lcosts", lc), ("bestlabels", bl)))
    df.nasumicni(); O.neighbour_proposals(p, d1, d2, pr, lc, npr, bl)
    st = df.host_state()
    ok2 = all(np.array_equal(st[k], v) for k, v in (("nprop", npr), ("proposals", pr), ("lcosts", lc)))
    df.ceoBCD(1); O.bcd_sweep(p, pr, lc, npr, bl)
    ok3 = np.array_equal(df.bestlabels.cpu().numpy(), bl)
    print("%3d  %4dx%-4d cells %2dx%-2d (%dx%d) window %d ngauss %2d  knn %s  neighbour %s  sweep %s" % (it, W, H, cw, ch, W // cw, H // ch, window, ngauss, ok, ok2, ok3), flush=True)
    bad += not (ok and ok2 and ok3)
print("failures:", bad)
sys.exit(1 if bad else 0)
