"""How many chains of a BCD phase see exactly the input (their own line of bestlabels) they saw one sweep earlier?"""
import sys, os, importlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H, W = 436, 1024
img1, img2, gt = synth.make_pair(H, W, seed=2022)
df = pl.DiscreteFlow(H, W, seed=99)
df.load_pair(torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()); df.generisi(); df.nasumicni(); df.pakovanje()
prev_in = {}
for sweep in range(1, 7):
    out = []
    for ph in range(4):
        before = df.bestlabels.cpu().numpy().copy()
        df.bcd_phase(ph)
        after = df.bestlabels.cpu().numpy()
        if ph == 0: lines = [before[:, x] for x in range(0, W, 2)]; lines_a = [after[:, x] for x in range(0, W, 2)]
        elif ph == 1: lines = [before[y, :] for y in range(0, H, 2)]; lines_a = [after[y, :] for y in range(0, H, 2)]
        elif ph == 2: lines = [before[:, x] for x in range(1, W, 2)]; lines_a = [after[:, x] for x in range(1, W, 2)]
        else: lines = [before[y, :] for y in range(1, H, 2)]; lines_a = [after[y, :] for y in range(1, H, 2)]
        same_in = np.mean([np.array_equal(a, b) for a, b in zip(lines, prev_in.get(ph, [None] * len(lines)))]) if ph in prev_in else float("nan")
        unchanged = np.mean([np.array_equal(a, b) for a, b in zip(lines, lines_a)])
        changed_px = np.mean(before != after)
        prev_in[ph] = lines
        out.append("ph%d same-input %.2f fixed-point %.2f px-changed %.4f" % (ph, same_in, unchanged, changed_px))
    print("sweep", sweep, " | ".join(out))
