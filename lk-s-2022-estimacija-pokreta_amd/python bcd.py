#!/usr/bin/env python3
"""Drop-in for the reference's second CLI (README.md:15-21, python bcd.py:13-17):

    python "python bcd.py" <idx <= 99> <backward 0|1> <bcd_times> [--device cuda:N]

Loads the files the first CLI wrote (ucitajSvePodatkeDoBCD, python bcd.py:67-81; always the "posle 00"
labels), runs bcd_times BCD sweeps on the GPU and, after every sweep, writes the reference's two .npy files
(python bcd.py:282-283) plus a Middlebury .flo of the same flow.  packedksets.npy is neither needed nor read.
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.basename(os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("picindex"); ap.add_argument("backward", choices=("0", "1")); ap.add_argument("bcd_times", type=int)
    ap.add_argument("--cell"); ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    pipeline = importlib.import_module(PKG + ".pipeline")
    flowio = importlib.import_module(PKG + ".flowio")
    idx = a.picindex if len(a.picindex) > 1 else "0" + a.picindex
    proposals = np.load(flowio.stage_name(idx, a.backward, "proposals_nakon_gausa"))
    lcosts = np.load(flowio.stage_name(idx, a.backward, "lcosts_nakon_gausa"))
    nprop = np.load(flowio.stage_name(idx, a.backward, "nprop"))
    bestlabels = np.load(flowio.labels_name(idx, a.backward, 0))
    pich, picw = nprop.shape
    cellh, cellw = (int(v) for v in a.cell.lower().split("x")) if a.cell else pipeline.default_cells(pich, picw)
    df = pipeline.DiscreteFlow(pich, picw, cellh, cellw, device=a.device)
    df.set_host_state(proposals, lcosts, nprop, bestlabels)

    def save(w):
        flow = df.vratiKonacniFlow().cpu().numpy().astype(np.float64)
        np.save(flowio.flow_name(idx, a.backward, w), flow)
        np.save(flowio.labels_name(idx, a.backward, w), df.bestlabels.cpu().numpy().astype(np.int64))
        flowio.write_flo(flowio.flow_name(idx, a.backward, w)[:-4] + ".flo", flow)
        print("uradjen bcd broj", w)

    df.ceoBCD(a.bcd_times, on_sweep=save)


if __name__ == "__main__":
    main()
