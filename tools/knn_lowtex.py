"""kNN stage kernel by kernel (dflow_knn_proposals_timed) on a dense and a low-texture pair, then the other stages of the
pass on the low-texture pair: python tools/knn_lowtex.py [variant.so]"""
import sys, os, json, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
_lib = importlib.import_module("lk-s-2022-estimacija-pokreta_amd._lib")
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"): _lib.LIB_PATH = os.path.join(ROOT, "tools", "prof_build", sys.argv[1])
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
df = pl.DiscreteFlow(H, W, seed=99)


def timed(fn):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for style in ("dense", "low_texture"):
    img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(0, 0), style=style)
    df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda())
    df.generisi_timed()
    ms, issued = df.generisi_timed()
    rec = {"style": style, "knn_ms": {k: round(v, 3) for k, v in ms.items()}, "knn_total_ms": round(sum(ms.values()), 3)}
    if hasattr(df, "knn_stats"):
        rec["stats"] = df.knn_stats()
    rec["neighbour_ms"] = round(timed(df.nasumicni), 3)
    rec["lists_ms"] = round(timed(df.pakovanje), 3)
    rec["bcd4_ms"] = round(timed(lambda: df.ceoBCD(4)), 3)
    print(json.dumps(rec), flush=True)
