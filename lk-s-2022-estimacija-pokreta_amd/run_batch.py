#!/usr/bin/env python3
"""Batch driver for BASELINE configs 3 and 4: forward + backward pass of every pair, sharded over the GPUs of one node
(pass p -> rank p mod world, README.md:40 of the reference), RCCL gather of the flow fields on rank 0, then the
forward/backward consistency check (postprocessing.py:123-135) per pair on rank 0.

    python run_batch.py --pairs 8 --bcd-times 4 [--size 436x1024] [--thresh 10] [--out DIR]
    python -m torch.distributed.run --nproc-per-node 8 run_batch.py --pairs 8 ...

Inputs are synthetic pairs (synth.make_pair, seed 1000*pair); outputs per pair in DIR: the reference's flow .npy names
for both directions, a .flo of the forward flow, sparse_field_<pair>.npy and parovi_<pair>.txt.
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
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--bcd-times", type=int, default=4)
    ap.add_argument("--size", default="436x1024")
    ap.add_argument("--thresh", type=float, default=10.0)     # README.md:65 of the reference
    ap.add_argument("--out", default=".")
    ap.add_argument("--group", type=int, default=4, help="passes of a rank whose BCD sweeps share their launches")
    ap.add_argument("--cell", default=None, help="cell size HxW (default: the geometry's usual cells)")
    ap.add_argument("--fp16-descriptors", action="store_true", help="DAISY values rounded to binary16 (BASELINE configs[4])")
    ap.add_argument("--time", action="store_true", help="run the passes twice and report the wall time of the second run")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    pipeline = importlib.import_module(PKG + ".pipeline")
    sharding = importlib.import_module(PKG + ".sharding")
    synth = importlib.import_module(PKG + ".synth")
    flowio = importlib.import_module(PKG + ".flowio")
    evaluate = importlib.import_module(PKG + ".evaluate")
    H, W = (int(v) for v in a.size.lower().split("x"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    passes = [(pair, backward) for pair in range(a.pairs) for backward in (0, 1)]
    # this rank's pairs are generated and uploaded BEFORE the passes run: the compute functions are GPU work only
    mine = sharding.assign_passes(len(passes), world, rank)
    images = {}
    for pair in sorted({passes[i][0] for i in mine}):
        img1, img2, _ = synth.make_pair(H, W, seed=synth.pair_seed(pair, 0))
        images[pair] = (torch.from_numpy(img1).to(dev), torch.from_numpy(img2).to(dev))
    # the passes of a rank (forward and backward runs, several pairs) are independent: their front ends run one after the
    # other, their BCD sweeps as batched launches (dflow_bcd_sweep_batch), `group` passes at a time
    group = max(1, min(a.group, len(mine)))
    flags = importlib.import_module(PKG + "._lib").FLAG_DESCR_F16 if a.fp16_descriptors else 0
    ch, cw = (int(v) for v in a.cell.lower().split("x")) if a.cell else (None, None)
    dfs = [pipeline.DiscreteFlow(H, W, ch, cw, device=dev, seed=rank, flags=flags) for _ in range(group)]

    front = [torch.cuda.Stream(device=dev) for _ in range(min(3, group))]     # front ends of a group's passes side by side

    def compute_many(descs):
        flows = []
        main = torch.cuda.current_stream(dev)
        for g0 in range(0, len(descs), group):
            part = descs[g0:g0 + group]
            start = torch.cuda.Event()
            start.record(main)                                   # the previous group's read-out is done
            done = []
            for j, (df, (pair, backward)) in enumerate(zip(dfs, part)):
                img1, img2 = images[pair]
                if backward:
                    img1, img2 = img2, img1
                st = front[j % len(front)]
                with torch.cuda.stream(st):
                    st.wait_event(start)
                    df.load_pair(img1, img2)
                    df.generisi()
                    df.nasumicni()
                    df.pakovanje()
                    e = torch.cuda.Event()
                    e.record(st)
                    done.append(e)
            for e in done:
                main.wait_event(e)
            pipeline.ceoBCD_batch(dfs[:len(part)], a.bcd_times)
            flows += [df.vratiKonacniFlow().clone() for df in dfs[:len(part)]]
        return flows

    flows = sharding.run_passes(passes, None, world, rank, dfs[0].flow, compute_many=compute_many)
    if a.time:
        import time
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        flows = sharding.run_passes(passes, None, world, rank, dfs[0].flow, compute_many=compute_many)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            print("%d passes (%d pairs, forward + backward) of %dx%d, bcd_times=%d on %d GPU(s): %.1f ms = %.2f ms per pass = %.1f Mpix/s"
                  % (len(passes), a.pairs, W, H, a.bcd_times, world, dt * 1e3, dt * 1e3 / len(passes), len(passes) * H * W / dt / 1e6))
    if rank == 0:
        os.makedirs(a.out, exist_ok=True)
        for pair in range(a.pairs):
            fwd, bwd = flows[2 * pair], flows[2 * pair + 1]
            sparse = pipeline.fb_consistency(fwd, bwd, a.thresh).cpu().numpy()
            for backward, f in ((0, fwd), (1, bwd)):
                np.save(os.path.join(a.out, flowio.flow_name(pair, backward, a.bcd_times)), f.cpu().numpy().astype(np.float64))
            flowio.write_flo(os.path.join(a.out, flowio.flow_name(pair, 0, a.bcd_times)[:-4] + ".flo"), fwd.cpu().numpy())
            np.save(os.path.join(a.out, "sparse_field_%02d.npy" % pair), sparse)
            evaluate.parovi(sparse, os.path.join(a.out, "parovi_%02d.txt" % pair))
            print("pair %d: %.1f%% of the forward flow survives the consistency check" % (pair, 100.0 * sparse[..., 2].mean()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
