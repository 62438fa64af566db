"""CPU checks of the synthetic inputs and of bench.py's accounting helpers (nothing here touches a GPU)."""
import hashlib
import json
import os
import sys

import numpy as np

from conftest import ROOT


def test_dense_style_is_the_bench_input_of_every_round(synth):
    # the bench pairs must not move when styles are added: digest of the first bench pair (seed 0) as of round 1
    img1, img2, gt = synth.make_pair(436, 1024, seed=synth.pair_seed(0, 0))
    assert hashlib.sha256(img1.tobytes() + img2.tobytes()).hexdigest()[:16] == "62ddf99d4237915a"
    assert gt.shape == (436, 1024, 2)


def test_low_texture_style(synth, oracle):
    H, W = 436, 1024
    a = synth.make_pair(H, W, seed=7, style="low_texture")
    b = synth.make_pair(H, W, seed=7, style="low_texture")
    assert all(np.array_equal(x, y) for x, y in zip(a, b))                      # deterministic
    reg = synth.low_texture_regions(H, W, 7)
    frac = [(reg == k).mean() for k in range(5)]
    assert frac[1] > 0.28 and frac[2] > 0.2 and frac[3] > 0.04 and frac[4] > 0.04 and frac[0] > 0.25
    img1, img2, gt = a
    sky = reg == 1
    assert (img1[sky] == 255).all()                                            # saturated in image 1 ...
    assert (img2 == 255).all(-1).mean() > 0.25                                 # ... and, after warp and noise, in image 2
    road = reg == 2
    assert img1[road].astype(int).max() - img1[road].astype(int).min() <= 6    # +-2 grey levels of texture (+ rounding)
    # what it is for: exactly-zero DAISY over the sky (daisy i flann.py:66: NRM_NONE), tiny descriptors on the road
    d1 = oracle.daisy(img1[:120, :256])
    assert (~d1.any(-1))[:60].mean() > 0.8
    try:
        synth.make_pair(H, W, style="nope")
        assert False
    except ValueError:
        pass


def test_bench_accounting_helpers():
    sys.path.insert(0, ROOT)
    import bench
    # interior pixels see 25 window cells; frames of 5 x 5 cells 3.8^2, of 2 x 2 cells 4
    assert abs(bench.mean_window_cells(125, 365, 25, 73) - 14.44) < 1e-9
    assert bench.mean_window_cells(54, 128, 27, 64) == 4.0
    assert 21.0 < bench.mean_window_cells(436, 1024, 27, 64) < 21.5
    # (query, candidate) pairs = pixels x candidates of their window: consistent with the mean window size
    assert abs(bench.knn_pairs(125, 365, 25, 73) / (125 * 365) / (25 * 73) - 14.44) < 1e-9
    d = bench.csrc_digest()
    assert len(d) == 16 and d == bench.csrc_digest()
    stale, detail = bench.profiles_state()
    assert detail["csrc_sha16_now"] == d and isinstance(stale, bool)
    # the tracked PMC summary carries what total_traffic_per_pass needs
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert "kernels" in t
    if "passes_profiled" in t:
        tot, top = bench.total_traffic_per_pass()
        assert tot > 1e9 and len(top) >= 3
