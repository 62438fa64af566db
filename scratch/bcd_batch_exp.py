"""BCD sweeps of B independent passes in ONE launch per phase (dflow_bcd_sweep_batch) vs B streams: ms per (pair, sweep)."""
import sys, os, importlib, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
BMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img1, img2, gt = synth.make_pair(H, W, seed=synth.pair_seed(0, 0))
a, b = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
dfs = []
for i in range(BMAX):
    df = pl.DiscreteFlow(H, W, seed=i)
    df.load_pair(a, b); df.generisi(); df.nasumicni(); df.pakovanje()
    dfs.append(df)
torch.cuda.synchronize()
def ev_time(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
print("pakovanje (lists kernel) %.3f ms" % ev_time(dfs[0].pakovanje))
for ph in range(4):
    print("phase %d alone %.3f ms" % (ph, ev_time(lambda: dfs[0].bcd_phase(ph))))
for B in range(1, BMAX + 1):
    t = ev_time(lambda: pl.ceoBCD_batch(dfs[:B], 2))
    print("batch B=%d  %.3f ms for 2 sweeps  ->  %.3f ms per (pair, sweep)" % (B, t, t / (2 * B)), flush=True)
streams = [torch.cuda.Stream() for _ in range(BMAX)]
for P in (2, 3, 4, 6):
    if P > BMAX: break
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                dfs[i].ceoBCD(2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("streams P=%d  wall %.3f ms for 2 sweeps  ->  %.3f ms per (pair, sweep)" % (P, dt, dt / (2 * P)), flush=True)
