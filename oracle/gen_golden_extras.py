#!/usr/bin/env python3
"""Golden vectors for the SURVEY 8(f) #3/#4 extras, produced by running the reference here (test infrastructure, like
gen_golden.py; only tests/ uses the output):

  * pakovanjeZaC (daisy i flann.py:321-398): `daisy i flann.py 6 0 0` under gen_golden's stubs on fixture a -> sha256 of
    the four 'pakovani za c k.npy' arrays.
  * removeSmallSegments (postprocessing.py:29-76): the reference function (plain import) on seeded random sparse fields.
  * FlowImage.readFlowFieldFromImage (visualization.py:37-53): the decoded field of a 16-bit PNG written by the
    build's writer, through the reference's text with a cv2 stub whose imread returns the decoded BGR array.

    python oracle/gen_golden_extras.py   ->  tests/golden/ref_extras.npz
"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import gen_golden as G          # noqa: E402
O = G.O


def main():
    store = {}
    # ---- pakovanjeZaC on fixture a (same seeds as gen_golden.make_fixture("a40x48_c5x6", ...), forward pass)
    H, W, ch, cw, seed = 40, 48, 5, 6, 11
    p = O.make_params(H, W, ch, cw, seed=seed)
    img1, img2, _ = G.synth.make_pair(H, W, seed=seed, amp_x=0.12 * W, amp_y=0.12 * H)
    with tempfile.TemporaryDirectory() as wd:
        ref = G.run_reference_pass(p, img1, img2, 6, 0, 0, wd, dopython=0)
    for k, a in enumerate(ref["za_c"]):
        store[f"za_c{k}_sha"] = G.digest(a, np.uint8)
        store[f"za_c{k}_shape"] = np.array(a.shape)
    store["za_c_nprop"] = ref["nprop"].astype(np.int16)
    # ---- removeSmallSegments
    sys.path.insert(0, G.REF)
    try:
        post = importlib.import_module("postprocessing")
    finally:
        sys.path.remove(G.REF)
    rng = np.random.default_rng(77)
    for i, (A, B, tresh, mins, pvalid) in enumerate(((24, 31, 2, 12, 0.8), (30, 22, 1, 6, 0.6), (17, 40, 3, 40, 0.9))):
        f = np.zeros((A, B, 3), np.float32)
        base = rng.integers(-3, 4, size=(A // 4 + 1, B // 4 + 1, 2)).astype(np.float32)
        f[..., :2] = np.kron(base, np.ones((4, 4, 1), np.float32))[:A, :B] + rng.integers(0, 2, size=(A, B, 2))
        f[..., 2] = rng.random((A, B)) < pvalid
        f[..., :2] *= f[..., 2:3]                       # invalid pixels carry (0,0) like consistencyCheck leaves them
        store[f"seg{i}_in"] = f.copy(); store[f"seg{i}_par"] = np.array([tresh, mins])
        post.removeSmallSegments(f, tresh, mins)
        store[f"seg{i}_out"] = f
        print("segments case", i, "valid before/after", int(store[f"seg{i}_in"][..., 2].sum()), int(f[..., 2].sum()))
    # ---- KITTI flow PNG through the reference's reader
    dio = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.flowio")
    uvv = np.zeros((9, 13, 3), np.float32)
    uvv[..., 0] = rng.integers(-2000, 2000, size=(9, 13)) / 64.0
    uvv[..., 1] = rng.integers(-900, 900, size=(9, 13)) / 64.0
    uvv[..., 2] = rng.random((9, 13)) < 0.7
    uvv[..., :2] *= uvv[..., 2:3]
    lines = open(os.path.join(G.REF, "visualization.py")).read().splitlines()
    with tempfile.TemporaryDirectory() as wd:
        path = os.path.join(wd, "gt.png")
        dio.write_kitti_flow_png(path, uvv)
        store["png_bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8)
        rgb16 = dio.read_png16(path)                    # (H,W,3) uint16 R,G,B: what cv2.imread(-1) returns, in BGR order
        cv2 = types.ModuleType("cv2")
        cv2.imread = lambda fn, flag: rgb16[..., ::-1].copy()
        cv2.COLOR_BGR2RGB = 4
        cv2.cvtColor = lambda img, code: img[..., ::-1].copy()
        ns = {"np": np, "os": os, "cv2": cv2, "cmap": None}
        exec("\n".join(lines[30:126]), ns)              # class FlowImage (visualization.py:31-126)
        fi = ns["FlowImage"](); fi.ucitajFlow(path)
        store["png_field_by_reference"] = fi.flow
    store["png_uvv"] = uvv
    out = os.path.join(ROOT, "tests", "golden", "ref_extras.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
