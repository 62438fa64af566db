"""The CPU oracle against vectors produced by the reference's own code (oracle/gen_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN_NAMES


def digest(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=dtype).tobytes()).hexdigest()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
@pytest.mark.parametrize("backward", (0, 1))
def test_pass_matches_reference(oracle, golden, name, backward):
    O, g = oracle, golden(name)
    p = O.make_params(int(g["H"]), int(g["W"]), int(g["cellh"]), int(g["cellw"]), seed=int(g["seed"]))
    k = "b%d_" % backward
    a, b = (g["img1"], g["img2"]) if backward == 0 else (g["img2"], g["img1"])
    d1, d2 = O.daisy(a), O.daisy(b)
    # descriptors were injected into the reference run; their digest guards against libm drift
    assert digest(d1, np.float32) == str(g[k + "d1_sha"]), "oracle DAISY differs from the build container's"
    assert digest(d2, np.float32) == str(g[k + "d2_sha"])
    # G1: generisi glue (slots, [dy,dx], truncated L1 in numpy order, WTA)
    pr, lc, npr, bl = O.knn_proposals(p, d1, d2)
    assert np.array_equal(bl, g[k + "labels00"])
    assert np.array_equal(O.labels_to_flow(p, pr, bl), g[k + "flow00"])
    # G2: nasumicni with replayed draws
    O.neighbour_proposals(p, d1, d2, pr, lc, npr, bl)
    assert np.array_equal(npr, g[k + "nprop"])
    assert np.array_equal(pr[:2], g[k + "proposals_rows"])
    assert np.array_equal(lc[:2], g[k + "lcosts_rows"])
    assert digest(pr, np.int64) == str(g[k + "proposals_sha"])
    assert digest(lc, np.float64) == str(g[k + "lcosts_sha"])
    # G3: pakovanje
    pk = O.pack_compat(p, pr, npr)
    assert np.array_equal(pk[1, 1], g[k + "packedksets_px"])
    assert digest(pk, np.uint8) == str(g[k + "packedksets_sha"])
    # G4: BCD sweeps
    for w in range(1, int(g["bcd_times"]) + 1):
        O.bcd_sweep(p, pr, lc, npr, bl)
        assert np.array_equal(bl, g[k + "labels%02d" % w]), "labels differ after sweep %d" % w
        assert digest(O.labels_to_flow(p, pr, bl), np.float64) == str(g[k + "flow%02d_sha" % w])


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_fb_consistency_matches_reference(oracle, golden, name):
    O, g = oracle, golden(name)
    p = O.make_params(int(g["H"]), int(g["W"]), int(g["cellh"]), int(g["cellw"]), seed=int(g["seed"]))
    flows = []
    for backward in (0, 1):
        a, b = (g["img1"], g["img2"]) if backward == 0 else (g["img2"], g["img1"])
        r = O.full_pass(p, a, b, int(g["bcd_times"]))
        flows.append(r["flows"][-1])
    for t in (1, 3):
        assert np.array_equal(O.fb_consistency(flows[0], flows[1], t), g["sparse_t%d" % t])


def test_fb_consistency_is_transposed(oracle):
    # a pure +3 px horizontal flow on a 6x9 field invalidates the last 3 ROWS
    fwd = np.zeros((6, 9, 2)); fwd[..., 1] = 3.0
    bwd = np.zeros((6, 9, 2)); bwd[..., 1] = -3.0
    s = oracle.fb_consistency(fwd, bwd, 10)
    assert s[:3, :, 2].all() and not s[3:, :, 2].any()


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_flo_roundtrip(golden, name):
    from conftest import pkg
    import os, tempfile
    g = golden(name)
    fio = pkg("flowio")
    parsed = g["flo_parsed_by_reference"]          # what visualization.py:9-29 read from our writer's bytes
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.flo")
        with open(path, "wb") as f:
            f.write(g["flo_bytes"].tobytes())
        assert np.array_equal(fio.read_flo(path), parsed)
        flow_dydx = parsed[..., ::-1]
        fio.write_flo(path, flow_dydx)
        assert open(path, "rb").read() == g["flo_bytes"].tobytes()


def test_knn_canonical_order(oracle):
    rng = np.random.default_rng(5)
    pts = rng.random((40, 68), dtype=np.float32)
    pts[7] = pts[3]; pts[20] = pts[3]                 # exact ties -> lower index first
    idx, dist = oracle.knn_points(pts[3], pts, 5)
    assert list(idx[:3]) == [3, 7, 20] and dist[0] == 0.0 and np.all(np.diff(dist) >= 0)
    d64 = ((pts.astype(np.float64) - pts[3].astype(np.float64)) ** 2).sum(1)
    assert set(idx) == set(np.argsort(d64, kind="stable")[:5])


def test_gauss_sampler_statistics(oracle):
    thr = oracle.gauss_thresholds(8.0)
    assert np.all(np.diff(thr.astype(np.int64)) >= 0)
    rng = np.random.default_rng(0)
    us = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64)
    offs = np.array([oracle.gauss_offset(thr, int(u)) for u in us])
    # floor(8 z): mean -0.5, std ~ sqrt(64 + 1/12)
    assert abs(offs.mean() + 0.5) < 0.25 and abs(offs.std() - 8.0) < 0.3
    # tails below 2^-32 cannot be drawn from a 32-bit uniform: the extreme offsets are about +-6.2 sigma
    assert -56 <= oracle.gauss_offset(thr, 0) <= -48 and 47 <= oracle.gauss_offset(thr, 2 ** 32 - 1) <= 63
    assert oracle.gauss_offset(thr, 2 ** 31) == 0 and oracle.gauss_offset(thr, 2 ** 31 - 1) == -1


def test_daisy_structure(oracle, synth):
    img, _, _ = synth.make_pair(40, 56, seed=3)
    d = oracle.daisy(img)
    assert d.shape == (40, 56, 68) and d.dtype == np.float32 and np.all(d >= 0) and np.isfinite(d).all()
    assert not d[-2:, :, :4].any() and not d[:, -2:, :4].any()   # reads within 2 px of the far edges are zeroed (centre bins)
    flat = np.full((40, 56, 3), 77, np.uint8)
    assert not oracle.daisy(flat).any()                       # no gradient -> zero descriptor
    # a horizontal ramp excites only the +x orientation bin (bin 0); the -x bin (2) stays zero
    ramp = np.repeat(np.arange(56, dtype=np.uint8)[None, :, None] * 4, 40, 0).repeat(3, 2)
    dr = oracle.daisy(np.ascontiguousarray(ramp))
    c = dr[20, 20].reshape(17, 4)
    assert np.all(c[:, 0] > 0) and np.all(c[:, 2] == 0)


def test_threaded_oracle_equals_serial(oracle, synth):
    """orc_set_threads only distributes independent units (rows, columns, chains): the results must not depend on it."""
    O = oracle
    H, W, ch, cw = 40, 48, 5, 6
    img1, img2, _ = synth.make_pair(H, W, seed=9, amp_x=4, amp_y=3)
    p = O.make_params(H, W, ch, cw, seed=2)
    try:
        O.set_threads(1)
        a = O.full_pass(p, img1, img2, 2)
        O.set_threads(4)
        assert O.get_threads() == 4
        b = O.full_pass(p, img1, img2, 2)
    finally:
        O.set_threads(1)
    for k in ("proposals", "lcosts", "nprop", "bestlabels"):
        assert np.array_equal(a[k], b[k]), k
    assert all(np.array_equal(x, y) for x, y in zip(a["flows"], b["flows"]))
