#!/usr/bin/env python3
"""Drop-in for the reference's first CLI (README.md:6-12, daisy i flann.py:16-27):

    python "daisy i flann.py" <idx <= 99> <backward 0|1> <dopython 0|1> [options]

Same positional arguments, same image naming (../data_scene_flow/training/image_2/0001{idx}_1{0|1}.png), same
output files in the current directory: the WTA flow/labels "posle 00", proposals_nakon_gausa,
lcosts_nakon_gausa, nprop -- in the reference's dtypes -- plus a Middlebury .flo next to every flow .npy.
The compat bit matrices (packedksets; with dopython=0 the four 'pakovani za c' copies) exist for users of the reference's own
BCD scripts (its `python bcd.py:76` cannot start without packedksets.npy); the GPU BCD of this package builds its own compact
lists in HBM and never reads them.  They are written by default while the file stays below PACKEDKSETS_DEFAULT_LIMIT bytes
(256 MiB: frames up to about 47 000 pixels); for larger frames (KITTI: 2.6 GB) one line says so and names the flag,
--packedksets writes them regardless of size, --no-packedksets never.
All computation runs in libdflow.so on the GPU; there is no CPU fallback.

Options for inputs the reference cannot handle: --image1/--image2 PATH, --cell HxW, --synthetic HxW
(synthetic pair with seed 1000*idx+backward), --seed N (neighbour-sampler key), --device cuda:N,
--fp16-descriptors (BASELINE configs[4]: DAISY values rounded to binary16).
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.basename(os.path.dirname(os.path.abspath(__file__)))


PACKEDKSETS_DEFAULT_LIMIT = 256 << 20


def read_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("picindex"); ap.add_argument("backward", choices=("0", "1")); ap.add_argument("dopython", choices=("0", "1"))
    ap.add_argument("--image1"); ap.add_argument("--image2"); ap.add_argument("--cell"); ap.add_argument("--synthetic")
    ap.add_argument("--seed", type=int, default=0); ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--packedksets", action="store_true", help="write the reference's compat-matrix file(s) whatever their size")
    ap.add_argument("--no-packedksets", action="store_true", help="never write them")
    ap.add_argument("--fp16-descriptors", action="store_true", help="round the DAISY values to binary16 (DFLOW_FLAG_DESCR_F16)")
    a = ap.parse_args(argv)
    pipeline = importlib.import_module(PKG + ".pipeline")
    flowio = importlib.import_module(PKG + ".flowio")
    synth = importlib.import_module(PKG + ".synth")
    idx = a.picindex if len(a.picindex) > 1 else "0" + a.picindex            # daisy i flann.py:24-25
    if a.synthetic:
        h, w = (int(v) for v in a.synthetic.lower().split("x"))
        pic1, pic2, _ = synth.make_pair(h, w, seed=synth.pair_seed(int(idx), 0))
        if a.backward == "1":
            pic1, pic2 = pic2, pic1
    else:
        other = "1" if a.backward == "0" else "0"                           # :19-22
        base = "../data_scene_flow/training/image_2/0001" + idx + "_1"
        pic1 = read_bgr(a.image1 or base + a.backward + ".png")
        pic2 = read_bgr(a.image2 or base + other + ".png")
        if not (a.image1 or a.image2):                                      # :34-35,52-53 KITTI crop
            pic1, pic2 = pic1[:375, :1241], pic2[:375, :1241]
    pich, picw = pic1.shape[:2]
    cellh, cellw = (int(v) for v in a.cell.lower().split("x")) if a.cell else pipeline.default_cells(pich, picw)
    flags = importlib.import_module(PKG + "._lib").FLAG_DESCR_F16 if a.fp16_descriptors else 0
    df = pipeline.DiscreteFlow(pich, picw, cellh, cellw, device=a.device, seed=a.seed, flags=flags)
    df.load_pair(np.ascontiguousarray(pic1), np.ascontiguousarray(pic2))    # :406-407
    df.generisi()                                                           # :409-412
    flow0 = df.vratiKonacniFlow().cpu().numpy().astype(np.float64)
    st0 = df.host_state()
    np.save(flowio.flow_name(idx, a.backward, 0), flow0)                    # sacuvajPodatke0 :200-202
    np.save(flowio.labels_name(idx, a.backward, 0), st0["bestlabels"])
    flowio.write_flo(flowio.flow_name(idx, a.backward, 0)[:-4] + ".flo", flow0)
    df.nasumicni()                                                          # :418
    st = df.host_state()
    np.save(flowio.stage_name(idx, a.backward, "proposals_nakon_gausa"), st["proposals"])   # sacuvajPodatke1 :249-253
    np.save(flowio.stage_name(idx, a.backward, "lcosts_nakon_gausa"), st["lcosts"])
    np.save(flowio.stage_name(idx, a.backward, "nprop"), st["nprop"])
    pk_bytes = pich * picw * 2 * (df.p.maxnprop * df.p.maxnprop // 8 + 1)
    if not a.packedksets and not a.no_packedksets and pk_bytes > PACKEDKSETS_DEFAULT_LIMIT:
        print("daisy i flann: packedksets (%.2f GB) not written: this package's `python bcd.py` builds its own lists on the GPU; "
              "pass --packedksets if the reference's own `python bcd.py` is to read these files" % (pk_bytes / 1e9))
    if a.packedksets or (not a.no_packedksets and pk_bytes <= PACKEDKSETS_DEFAULT_LIMIT):   # pakovanje :308 / pakovanjeZaC :394-397
        compat = importlib.import_module(PKG + ".compat")
        pk = compat.packedksets(df)
        if a.dopython == "1":
            np.save(flowio.stage_name(idx, a.backward, "packedksets"), pk)
        else:
            for k, arr in enumerate(compat.pakovani_za_c(pk)):
                np.save(flowio.stage_name(idx, a.backward, "pakovani za c %d" % k), arr)
    print("daisy i flann: %dx%d, cells %dx%d, nprop %d..%d" % (picw, pich, cellw, cellh, st["nprop"].min(), st["nprop"].max()))


if __name__ == "__main__":
    main()
