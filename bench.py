#!/usr/bin/env python3
"""bench.py -- throughput of the dense discrete optical-flow hot path on MI355X.

Metric (BASELINE.json): Mpix/s of flow at 1024x436 (Sintel shape), bcd_times=4.
A "step" is one full pass of the hot path over one synthetic image pair that is already resident in HBM:
DAISY x2 -> per-cell exact 5-NN proposals -> neighbour proposals -> 4 BCD sweeps -> labels->flow.
N > 1: one process per GPU (torch.distributed over RCCL); every rank processes its own pairs (weak scaling,
no data-path collective) and the flow fields are gathered on rank 0 over xGMI after every step.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "lk-s-2022-estimacija-pokreta_amd"

H, W, BCD_TIMES = 436, 1024, 4
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense (the kNN screen runs on f16 MFMA)


def knn_pairs(pich, picw, cellh, cellw, window=2):
    """Number of (image-1 pixel, image-2 candidate) distance evaluations of one pass."""
    ncx, ncy = picw // cellw, pich // cellh
    wx = [(picw if c == ncx - 1 else (c + 1) * cellw) - c * cellw for c in range(ncx)]
    wy = [(pich if c == ncy - 1 else (c + 1) * cellh) - c * cellh for c in range(ncy)]
    sx = sum(wx[ci] * sum(wx[max(0, ci - window):ci + window + 1]) for ci in range(ncx))
    sy = sum(wy[cj] * sum(wy[max(0, cj - window):cj + window + 1]) for cj in range(ncy))
    return sx * sy


def cpu_baseline(synth):
    """The CPU oracle (a C port of the reference path, single thread) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    h, w = 109, 512                                  # 1/8 of the frame area, same 27x64 cells, same bcd_times
    img1, img2, _ = synth.make_pair(h, w, seed=4242, amp_x=40.0, amp_y=20.0)
    p = O.make_params(h, w, 27, 64, seed=1)
    t0 = time.perf_counter()
    O.full_pass(p, img1, img2, BCD_TIMES)
    dt = time.perf_counter() - t0
    return {"value": h * w / dt / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "%dx%d crop-sized synthetic pair (1/8 frame), cells 27x64, bcd_times=%d, %.1f s; "
                      "C restatement of daisy i flann.py + python bcd.py (oracle/), exact kNN instead of FLANN"
                      % (w, h, BCD_TIMES, dt)}


def stage_rooflines(torch, df, pair, cellh, cellw, reps=3):
    """the three per-stage fractions and the end-to-end one, from one pair at a time on one stream
    (no overlap with other pairs, unlike the timed region), HIP events between the stages, mean of `reps` passes."""
    names = ["daisy", "knn", "neighbour", "pakovanje", "bcd", "labels_to_flow"]
    acc = dict.fromkeys(names, 0.0)
    st = torch.cuda.Stream(device=df.device)
    for _ in range(reps):
        with torch.cuda.stream(st):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
            evs[0].record()
            df.load_pair(*pair); evs[1].record()
            df.generisi(); evs[2].record()
            df.nasumicni(); evs[3].record()
            df.pakovanje(); evs[4].record()
            df.ceoBCD(BCD_TIMES); evs[5].record()
            df.vratiKonacniFlow(); evs[6].record()
        st.synchronize()
        for k, n in enumerate(names):
            acc[n] += evs[k].elapsed_time(evs[k + 1]) / reps
    N = H * W
    daisy_bytes = 2 * N * (3 + 68 * 4)                               # 550 B/px/pass
    knn_flops = 2 * 68 * knn_pairs(H, W, cellh, cellw)               # dense contraction of the +-2-cell window
    knn_bytes = 2 * N * 272 + N * 125 * 8
    bcd_bytes = BCD_TIMES * 2 * N * (150 * 4 + 150 * 4 + 16)         # 2.4 kB/px/sweep
    total_ms = sum(acc.values())
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9
    return {
        "measured": "one pair at a time on one stream, mean of %d passes" % reps,
        "ms": {n: round(acc[n], 4) for n in names}, "ms_total": round(total_ms, 4),
        "daisy": {"bound": "hbm", "bytes": daisy_bytes, "achieved": gbs(daisy_bytes, acc["daisy"]), "unit": "GB/s",
                  "peak": HBM_PEAK_GBS, "frac": gbs(daisy_bytes, acc["daisy"]) / HBM_PEAK_GBS},
        "knn": {"bound": "mfma", "flops": knn_flops, "achieved": knn_flops / (acc["knn"] * 1e-3) / 1e12,
                "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TFLOPS,
                "frac": knn_flops / (acc["knn"] * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                "note": "f16 MFMA screen with a rigorous error bound + exact f32 re-ranking of the survivors; "
                        "flops are the algorithmic 2*68 per (query, candidate) pair"},
        "bcd": {"bound": "hbm", "bytes": bcd_bytes, "achieved": gbs(bcd_bytes, acc["bcd"]), "unit": "GB/s",
                "peak": HBM_PEAK_GBS, "frac": gbs(bcd_bytes, acc["bcd"]) / HBM_PEAK_GBS},
        "end_to_end": {"bound": "hbm", "bytes": daisy_bytes + knn_bytes + bcd_bytes,
                       "achieved": gbs(daisy_bytes + knn_bytes + bcd_bytes, total_ms), "unit": "GB/s",
                       "peak": HBM_PEAK_GBS, "frac": gbs(daisy_bytes + knn_bytes + bcd_bytes, total_ms) / HBM_PEAK_GBS},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent pairs in flight per GPU (each on its own HIP stream and workspace)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    synth = importlib.import_module(PKG + ".synth")
    pipeline = importlib.import_module(PKG + ".pipeline")
    sharding = importlib.import_module(PKG + ".sharding")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run the process group is always created (also for one rank, which exercises the same
    # RCCL gather path); a plain `python bench.py` runs without torch.distributed
    use_dist = "RANK" in os.environ
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    cellh, cellw = pipeline.default_cells(H, W)
    # `inflight` independent pipelines per GPU: consecutive steps (= different image pairs) run on different HIP streams,
    # so the latency-bound BCD chains of one pair overlap the MFMA-bound kNN screening of the next.  Every step is still
    # one complete pass over one pair; the timed region contains exactly `steps` of them.
    P = max(1, min(args.inflight, args.steps))
    flows = [pipeline.DiscreteFlow(H, W, cellh, cellw, device=dev, seed=rank) for _ in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    # two distinct synthetic pairs per rank, resident in HBM before the timed region
    pairs = []
    for j in range(2):
        img1, img2, _ = synth.make_pair(H, W, seed=synth.pair_seed(2 * rank + j, 0))
        pairs.append((torch.from_numpy(img1).to(dev), torch.from_numpy(img2).to(dev)))
    gather_bufs = [sharding.make_gather_buffers(f.flow, world, rank) for f in flows] if use_dist else None

    ev = lambda: torch.cuda.Event(enable_timing=True)
    bcd_events = []

    def step(i, timed):
        df, st = flows[i % P], streams[i % P]
        a, b = pairs[i % 2]
        with torch.cuda.stream(st):
            df.load_pair(a, b)
            df.generisi()
            df.nasumicni()
            df.pakovanje()
            if timed:
                e0, e1 = ev(), ev()
                e0.record()
            df.ceoBCD(BCD_TIMES)
            if timed:
                e1.record()
                bcd_events.append((e0, e1))
            flow = df.vratiKonacniFlow()
            if use_dist:
                sharding.gather_flows(flow, gather_bufs[i % P], rank)
        return flow

    for i in range(args.warmup):
        step(i, False)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    sync_all()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # dominant kernel of a step: bcd_chain_kernel, 4 sweeps x 4 phases = 16 launches between the two events
        launches = 4 * BCD_TIMES
        bcd_ms = sum(a.elapsed_time(b) for a, b in bcd_events) / max(1, len(bcd_events)) / launches
        # algorithmic bytes of one phase launch: every pixel of half the image lines is visited once and
        # needs its labels: L*4 B flows + L*4 B costs + 16 B per pixel, L = 150  ->  1216 B per visited pixel
        alg_bytes = (H * W // 2) * (150 * 4 + 150 * 4 + 16)
        achieved = alg_bytes / (bcd_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpix/s flow (1024x436, bcd_times=4)",
            "value": world * args.steps * H * W / dt / 1e6,
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "single 1024x436 Sintel-shape pair per step per GPU, forward only, bcd_times=4 "
                                   "(BASELINE.json configs[1]); cells 64x27, 150 labels/px",
                       "pairs_in_flight_per_gpu": P,
                       "parallelism": "one pass per step; %d independent steps in flight per GPU on separate HIP streams; "
                                      "flow fields gathered on rank 0" % P},
            "roofline": {"bound": "hbm", "kernel": "bcd_chain_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # HBM-side bytes per launch from rocprofv3 --pmc (one counter per pass, profiles/r01_pmc_traffic.txt):
                         # FETCH_SIZE 598299 KB, doubled as MI355X_MICROARCH.md prescribes for gfx950's wide (16 B/lane) reads,
                         # + WRITE_SIZE 39138 KB.  Expected from the access pattern: 1.14 GB of 32-byte label records (160 rows
                         # per visited pixel: the compat lists the reference keeps in packedksets ride along) + 36 MB back-pointers
                         "traffic": (2 * 598299 + 39138) * 1024,
                         "launch_ms": bcd_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "serial Viterbi chains (218-512 workgroups x 436-1024 dependent steps): latency-bound, "
                                 "not bandwidth-bound; launch_ms is measured with %d pairs in flight" % P},
        }
        out["roofline"]["stages"] = stage_rooflines(torch, flows[0], pairs[0], cellh, cellw)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(synth)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
