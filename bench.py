#!/usr/bin/env python3
"""bench.py -- throughput of the dense discrete optical-flow hot path on MI355X.

Metric (BASELINE.json): Mpix/s of flow at 1024x436 (Sintel shape), bcd_times=4.
A "step" is one full pass of the hot path over one synthetic image pair that is already resident in HBM:
DAISY x2 -> per-cell exact 5-NN proposals -> neighbour proposals -> 4 BCD sweeps -> labels->flow.
N > 1: one process per GPU (torch.distributed over RCCL); every rank processes its own pairs (weak scaling,
no data-path collective) and the flow fields are gathered on rank 0 over xGMI after every step.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`python bench.py --gpus N` (N > 1, no RANK in the environment) starts its N ranks itself -- before anything touches the
GPU -- and relays rank 0's JSON line; under `python -m torch.distributed.run ... bench.py --gpus N` the ranks are the
launcher's.  Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "lk-s-2022-estimacija-pokreta_amd"

H, W, BCD_TIMES = 436, 1024, 4
LONG_STEPS = 98                # the default --steps; a run with another --steps adds one timed region of this length
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense (the kNN screen runs on f16 MFMA)
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # rocprofv3 --pmc passes of this command (tracked)


# ---------------------------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """One child process per GPU, started BEFORE this process has touched torch.cuda / HIP (a process that has
    initialised the GPU must never be replaced or forked into workers).  The children are plain `python bench.py ...`
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what torch.distributed.run
    would give them.  Rank 0's stdout (the JSON line) is relayed; the exit code is non-zero if any rank failed."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves the others waiting in a collective: stop them (exactly the processes started above)
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.1)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    for line in out0.splitlines():                 # the JSON line goes to stdout, library chatter (gloo) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------- accounting
def knn_pairs(pich, picw, cellh, cellw, window=2):
    """Number of (image-1 pixel, image-2 candidate) distance evaluations of one pass."""
    ncx, ncy = picw // cellw, pich // cellh
    wx = [(picw if c == ncx - 1 else (c + 1) * cellw) - c * cellw for c in range(ncx)]
    wy = [(pich if c == ncy - 1 else (c + 1) * cellh) - c * cellh for c in range(ncy)]
    sx = sum(wx[ci] * sum(wx[max(0, ci - window):ci + window + 1]) for ci in range(ncx))
    sy = sum(wy[cj] * sum(wy[max(0, cj - window):cj + window + 1]) for cj in range(ncy))
    return sx * sy


def mean_window_cells(pich, picw, cellh, cellw, window=2):
    """Mean number of cells in a pixel's search window (25 in the interior, fewer at the border and on small frames): the
    CPU cost per pixel is proportional to it, so Mpix/s of frames with different values do not compare directly."""
    ncx, ncy = picw // cellw, pich // cellh
    wx = [(picw if c == ncx - 1 else (c + 1) * cellw) - c * cellw for c in range(ncx)]
    wy = [(pich if c == ncy - 1 else (c + 1) * cellh) - c * cellh for c in range(ncy)]
    sx = sum(wx[c] * (min(ncx - 1, c + window) - max(0, c - window) + 1) for c in range(ncx))
    sy = sum(wy[c] * (min(ncy - 1, c + window) - max(0, c - window) + 1) for c in range(ncy))
    return sx * sy / float(pich * picw)


def epe_stats(flow, gt, mask=None):
    """Mean / median end-point error and % > 3 px (metric of visualization.py:128-152) of a (H,W,2) [dy,dx] field."""
    import numpy as np
    e = np.sqrt(((flow.astype(np.float64) - gt) ** 2).sum(-1))
    if mask is not None:
        e = e[mask]
    return {"mean": float(e.mean()), "median": float(np.median(e)), "pct_gt3": float((e > 3.0).mean() * 100.0),
            "pixels": int(e.size)}


def window_mask(gt, cellh, cellw, window=2):
    """Pixels whose ground-truth target lies inside the image AND inside the +-window-cell search window
    (daisy i flann.py:167-168): only there can a proposal reach the ground truth at all."""
    import numpy as np
    Hh, Ww, _ = gt.shape
    ncx, ncy = Ww // cellw, Hh // cellh
    yy, xx = np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij")
    ty = np.rint(yy + gt[..., 0]).astype(np.int64)
    tx = np.rint(xx + gt[..., 1]).astype(np.int64)
    inside = (ty >= 0) & (ty < Hh) & (tx >= 0) & (tx < Ww)
    cy = np.minimum(yy // cellh, ncy - 1); cx = np.minimum(xx // cellw, ncx - 1)
    tcy = np.minimum(np.clip(ty, 0, Hh - 1) // cellh, ncy - 1); tcx = np.minimum(np.clip(tx, 0, Ww - 1) // cellw, ncx - 1)
    return inside & (np.abs(tcy - cy) <= window) & (np.abs(tcx - cx) <= window)


def cpu_baseline(synth, gpu_flow, bench_seed, cellh, cellw):
    """The CPU oracle (a C port of the reference path) beside the GPU number, on this host's cores; about 15 s in all.

    Entry 1 (headline): the bench's own pair (BASELINE configs[1]: 1024x436, forward, bcd_times=4), the box's CPU share;
    its flow is also the EPE reference (`epe_delta_vs_oracle`).  Entry 2: BASELINE configs[0]'s geometry (cells 73x25 of
    1241x375, forward, 1 sweep: daisy i flann.py:34-35,42-43, python bcd.py:261-284) on a 5 x 5-cell part of such a frame.
    Entry 3: 2 x 2 cells of a Sintel frame single-threaded, the form the reference itself runs in (one core)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    host_cores = os.cpu_count()
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    threads = max(1, min(16, usable))          # a one-GPU box's CPU share
    entries = []
    O.set_threads(threads)
    img1, img2, gt = synth.make_pair(H, W, seed=bench_seed)
    t0 = time.perf_counter()
    ref = O.full_pass(O.make_params(H, W, cellh, cellw, seed=0), img1, img2, BCD_TIMES)
    dt = time.perf_counter() - t0
    oracle_flow = ref["flows"][-1]
    entries.append({"config": "BASELINE configs[1]: the bench pair, 1024x436, cells %dx%d, forward, bcd_times=%d" % (cellw, cellh, BCD_TIMES),
                    "value": H * W / dt / 1e6, "unit": "Mpix/s", "threads": threads, "seconds": dt,
                    "window_cells_per_pixel": round(mean_window_cells(H, W, cellh, cellw), 2)})
    kh, kw = 125, 365
    a, b, _ = synth.make_pair(kh, kw, seed=synth.pair_seed(6, 0), amp_x=30.0, amp_y=10.0)
    t0 = time.perf_counter()
    O.full_pass(O.make_params(kh, kw, 25, 73, seed=0), a, b, 1)
    dk = time.perf_counter() - t0
    entries.append({"config": "BASELINE configs[0] geometry: cells 73x25 (of 1241x375), idx 6 forward, bcd_times=1, on a 365x125 "
                              "part (5 x 5 cells) of a synthetic pair (KITTI is absent)",
                    "value": kh * kw / dk / 1e6, "unit": "Mpix/s", "threads": threads, "seconds": dk,
                    "window_cells_per_pixel": round(mean_window_cells(kh, kw, 25, 73), 2),
                    "window_cells_per_pixel_full_frame": round(mean_window_cells(375, 1241, 25, 73), 2)})
    O.set_threads(1)
    sh, sw = 54, 128
    a, b, _ = synth.make_pair(sh, sw, seed=4242, amp_x=12.0, amp_y=6.0)
    t0 = time.perf_counter()
    O.full_pass(O.make_params(sh, sw, 27, 64, seed=1), a, b, BCD_TIMES)
    ds = time.perf_counter() - t0
    entries.append({"config": "128x54 synthetic pair (2 x 2 cells of a Sintel frame: 4 window cells per pixel instead of up to 25), "
                              "cells 64x27, bcd_times=%d, single thread like the reference" % BCD_TIMES,
                    "value": sh * sw / ds / 1e6, "unit": "Mpix/s", "threads": 1, "seconds": ds,
                    "window_cells_per_pixel": round(mean_window_cells(sh, sw, 27, 64), 2),
                    "window_cells_per_pixel_full_frame": round(mean_window_cells(H, W, 27, 64), 2)})
    base = {"value": entries[0]["value"], "unit": "Mpix/s", "cores": threads, "kind": "port",
            "host_cores": host_cores, "usable_cores": usable,
            "sample": "the bench's own 1024x436 pair (whole frame, bcd_times=%d) in %.1f s on %d threads; C restatement of "
                      "daisy i flann.py + python bcd.py (oracle/), exact kNN instead of FLANN; the reference's cv2/FLANN "
                      "native code is absent and cannot be timed" % (BCD_TIMES, dt, threads),
            "entries_note": "Mpix/s of the entries are not comparable with each other: the work per pixel grows with window_cells_per_pixel "
                            "(entries 2 and 3 run on sub-frames with fewer window cells than their full frames; the bias favours the CPU)",
            "entries": entries}
    # EPE of both paths against the synthetic ground truth, and their difference (labels are bit-identical => 0)
    m = window_mask(gt, cellh, cellw)
    g_all, g_win = epe_stats(gpu_flow, gt), epe_stats(gpu_flow, gt, m)
    o_all, o_win = epe_stats(oracle_flow, gt), epe_stats(oracle_flow, gt, m)
    epe = {"gpu": {"all_pixels": g_all, "gt_inside_image_and_search_window": g_win},
           "oracle": {"all_pixels": o_all, "gt_inside_image_and_search_window": o_win},
           "flow_fields_identical": bool(np.array_equal(gpu_flow.astype(np.float64), oracle_flow)),
           "note": "DAISY is this build's restatement of OpenCV-contrib (parity unpinned: cv2 is absent), so EPE is vs the "
                   "restated pipeline, not vs cv2-DAISY + FLANN"}
    return base, epe, abs(g_all["mean"] - o_all["mean"])


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


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources: the tracked PMC summaries carry the digest they were collected with."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, PKG, "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")) or n == "Makefile":
            with open(os.path.join(d, n), "rb") as f:
                h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def profiles_state():
    """(stale, detail): whether profiles/pmc_traffic.json and pmc_knn.json were collected with the kernels that are running now."""
    now, detail, stale = csrc_digest(), {}, False
    for n in ("pmc_traffic.json", "pmc_knn.json"):
        try:
            with open(os.path.join(ROOT, "profiles", n)) as f:
                was = json.load(f).get("csrc_sha16")
        except Exception:
            was = None
        detail[n] = was
        stale |= was != now
    detail["csrc_sha16_now"] = now
    return stale, detail


def total_traffic_per_pass():
    """HBM bytes of ONE pass summed over every kernel of the profiled bench command (profiles/pmc_traffic.json: bytes per
    launch x launches / passes profiled), next to the algorithmic bytes: what the whole pipeline moves, not just one kernel."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            d = json.load(f)
        n = float(d["passes_profiled"])
        per = {k.split("(")[0]: v["hbm_bytes_per_launch"] * v["launches_in_pass"] / n for k, v in d["kernels"].items()}
        top = dict(sorted(per.items(), key=lambda kv: -kv[1])[:8])
        return int(sum(per.values())), {k: int(v) for k, v in top.items()}
    except Exception:
        return None, None


def pmc_traffic(kernel, unit_grids=None, passes=1.0):
    """HBM bytes per launch of `kernel` from the tracked rocprofv3 --pmc summary (profiles/pmc_traffic.json, written by
    tools/pmc_to_json.py from separate FETCH_SIZE and WRITE_SIZE passes of the bench command; FETCH_SIZE already doubled as
    MI355X_MICROARCH.md prescribes for gfx950's wide reads).  `unit_grids` = launch sizes in threads of ONE pass (column and
    row phases); a batched launch carries `passes` of them and moves that many times the bytes (every pass streams its own
    planes), so the per-pass traffic of every profiled multiple of a unit grid is averaged and scaled.  None if the file has
    no matching entry."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            d = json.load(f)
        e = d["kernels"][kernel]
        if not unit_grids:
            return int(e["hbm_bytes_per_launch"]), d.get("source", PMC_TRAFFIC_FILE)
        per_unit = []
        for u in unit_grids:
            v = [(g["hbm_bytes_per_launch"] / (int(k) // u), g["launches_in_pass"]) for k, g in e["by_grid_threads"].items()
                 if int(k) % u == 0 and int(k) // u >= 1]
            per_unit.append(sum(a * n for a, n in v) / sum(n for _, n in v))
        return int(passes * sum(per_unit) / len(per_unit)), d.get("source", PMC_TRAFFIC_FILE) + " (per pass, scaled to the passes of a launch)"
    except Exception:
        return None, None


# ---------------------------------------------------------------------------------------------------- engines
def _device_and_backend(args, local_rank):
    """Normally rank r drives GPU r over RCCL ("nccl").  --rehearse-on-one-gpu puts every rank on device 0 and gathers
    over gloo: RCCL cannot host two ranks on one device, and the development box has one GPU; the launcher, the engines and
    the torch.distributed calls are the same (tests/test_gpu_parity.py::test_bench_two_ranks_rehearsal)."""
    if args.rehearse_on_one_gpu:
        return 0, "gloo"
    return local_rank, "nccl"


def _make_jobs(eng, passes):
    """(image 1, image 2) device tensors of the passes p = 2 * pair + direction (synthetic pair 1000 * pair, the backward
    pass swaps the images), uploaded once per pair."""
    imgs, jobs = {}, []
    for p in passes:
        q, backward = p // 2, p % 2
        if q not in imgs:
            i1, i2, _ = eng.synth.make_pair(H, W, seed=eng.synth.pair_seed(q, 0))
            imgs[q] = (eng.torch.from_numpy(i1).to(eng.dev), eng.torch.from_numpy(i2).to(eng.dev))
        a, b = imgs[q]
        jobs.append((b, a) if backward else (a, b))
    return jobs


def _consistency(eng, fwd, bwd):
    return eng.pipeline.fb_consistency(fwd, bwd, 10.0, eng.flows[0].p)


class GpuEngine:
    """`inflight` independent pipelines per GPU: consecutive steps (= different image pairs) run on different HIP
    streams, so the latency-bound BCD chains of one pair overlap the MFMA-bound kNN screening of the next.  Every step is
    still one complete pass over one pair; the timed region contains exactly `steps` of them."""

    make_jobs, consistency = _make_jobs, _consistency

    def __init__(self, args, rank, local_rank, world):
        import torch
        self.torch = torch
        self.synth = importlib.import_module(PKG + ".synth")
        self.pipeline = importlib.import_module(PKG + ".pipeline")
        local_rank, self.backend = _device_and_backend(args, local_rank)
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.rank = rank
        self.cellh, self.cellw = self.pipeline.default_cells(H, W)
        self.P = max(1, min(args.inflight, args.steps))
        self.flows = [self.pipeline.DiscreteFlow(H, W, self.cellh, self.cellw, device=self.dev, seed=rank) for _ in range(self.P)]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.P)]
        # two distinct synthetic pairs per rank, resident in HBM before the timed region
        self.seeds = [self.synth.pair_seed(2 * rank + j, 0) for j in range(2)]
        self.pairs = []
        for sd in self.seeds:
            img1, img2, _ = self.synth.make_pair(H, W, seed=sd)
            self.pairs.append((torch.from_numpy(img1).to(self.dev), torch.from_numpy(img2).to(self.dev)))
        self.bcd_events = []
        self.gather = None
        self.jobs = None

    def like(self):
        return self.flows[0].flow

    def step(self, i, timed):
        torch = self.torch
        df, st = self.flows[i % self.P], self.streams[i % self.P]
        a, b = self.jobs[i] if self.jobs is not None else self.pairs[i % 2]
        with torch.cuda.stream(st):
            df.load_pair(a, b)
            df.generisi()
            df.nasumicni()
            df.pakovanje()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            df.ceoBCD(BCD_TIMES)
            if timed:
                e1.record()
                self.bcd_events.append((e0, e1, 1))
            flow = df.vratiKonacniFlow()
            if self.gather is not None:
                self.gather(flow, i % self.P)
        return flow

    def begin(self, nsteps):
        pass

    def sync(self):
        self.torch.cuda.synchronize()


class BatchEngine:
    """Steps are processed in groups of `batch` pairs: the front end of every pair of a group (DAISY, kNN proposals,
    neighbour proposals) runs on one of `front` HIP streams, the compat lists of a pair follow on a stream of their own as
    soon as that pair is ready (the front end paces the pipeline: 10.96 vs 11.07 ms per step with the lists off its streams),
    then the BCD sweeps of the whole group run on the BCD stream as ONE batched launch per phase (dflow_bcd_sweep_batch: chains x passes in one grid), then labels -> flow and the gather.
    Two sets of per-pair state alternate, so the front end of group g+1 overlaps the sweeps of group g.  Every step is
    still one complete pass over one pair; the timed region contains exactly `steps` of them (the last group may be
    smaller)."""

    make_jobs, consistency = _make_jobs, _consistency

    def __init__(self, args, rank, local_rank, world):
        import torch
        self.torch = torch
        self.synth = importlib.import_module(PKG + ".synth")
        self.pipeline = importlib.import_module(PKG + ".pipeline")
        local_rank, self.backend = _device_and_backend(args, local_rank)
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.rank = rank
        self.cellh, self.cellw = self.pipeline.default_cells(H, W)
        self.B = max(1, args.batch)
        self.P = self.B
        self.groups = [int(x) for x in args.groups.split(",")] if getattr(args, "groups", None) else None
        if self.groups and max(self.groups) > self.B:
            raise SystemExit("--groups: a group cannot exceed --batch")
        self.sets = [[self.pipeline.DiscreteFlow(H, W, self.cellh, self.cellw, device=self.dev, seed=rank) for _ in range(self.B)]
                     for _ in range(2)]
        self.flows = self.sets[0]
        self.front = [torch.cuda.Stream(device=self.dev) for _ in range(max(1, args.front))]
        self.bcd_stream = torch.cuda.Stream(device=self.dev)
        self.lists_on_bcd = not bool(getattr(args, "lists_on_front", False))
        # the lists of a pair start as soon as that pair's front end is done (a stream of their own), not when the whole group is
        self.lists_stream = torch.cuda.Stream(device=self.dev) if self.lists_on_bcd and not getattr(args, "no_lists_stream", False) else None
        self.set_free = [None, None]            # event: the set's previous sweeps + flow read-out are done
        self.seeds = [self.synth.pair_seed(2 * rank + j, 0) for j in range(2)]
        self.pairs = []
        for sd in self.seeds:
            img1, img2, _ = self.synth.make_pair(H, W, seed=sd)
            self.pairs.append((torch.from_numpy(img1).to(self.dev), torch.from_numpy(img2).to(self.dev)))
        self.bcd_events = []
        self.gather = None
        self.pending = []                       # step indices of the group being collected
        self.group_no = 0
        self.nsteps = None
        self.jobs = None                        # explicit (image 1, image 2) per step (the fixed batch); None: the two resident pairs in turn

    def like(self):
        return self.flows[0].flow

    def begin(self, nsteps):
        self.pending, self.nsteps, self.done = [], nsteps, 0
        if self.groups and sum(self.groups) == nsteps:
            self.plan = list(self.groups)
        else:
            ngroups = max(1, -(-nsteps // self.B))          # groups of equal size, at most `batch` pairs each
            q, r = divmod(nsteps, ngroups)
            # the smaller groups first: the first group's front end has no sweeps to overlap with (20 steps as 6, 7, 7: 11.06 ms
            # per step; as 7, 7, 6: 11.23)
            self.plan = [q] * (ngroups - r) + [q + 1] * r
        self.plan_at = 0

    def step(self, i, timed):
        self.pending.append(i)
        self.done += 1
        if len(self.pending) == self.plan[self.plan_at] or self.done == self.nsteps:
            self._run_group(self.pending, timed)
            self.pending = []
            self.plan_at = min(self.plan_at + 1, len(self.plan) - 1)

    def _run_group(self, idx, timed):
        torch = self.torch
        k = self.group_no % 2
        self.group_no += 1
        dfs = self.sets[k][:len(idx)]
        evs = []
        for j, i in enumerate(idx):
            st = self.front[j % len(self.front)]
            a, b = self.jobs[i] if self.jobs is not None else self.pairs[i % 2]
            with torch.cuda.stream(st):
                if self.set_free[k] is not None:
                    st.wait_event(self.set_free[k])
                df = dfs[j]
                df.load_pair(a, b)
                df.generisi()
                df.nasumicni()
                if not self.lists_on_bcd:
                    df.pakovanje()
                e = torch.cuda.Event()
                e.record()
                evs.append(e)
        if self.lists_stream is not None:
            evs2 = []
            with torch.cuda.stream(self.lists_stream):
                for e, df in zip(evs, dfs):
                    self.lists_stream.wait_event(e)
                    df.pakovanje()
                    e2 = torch.cuda.Event()
                    e2.record()
                    evs2.append(e2)
            evs = evs2
        with torch.cuda.stream(self.bcd_stream):
            for e in evs:
                self.bcd_stream.wait_event(e)
            if self.lists_on_bcd and self.lists_stream is None:
                for df in dfs:
                    df.pakovanje()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.pipeline.ceoBCD_batch(dfs, BCD_TIMES)
            if timed:
                e1.record()
                self.bcd_events.append((e0, e1, len(idx)))
            for j, df in enumerate(dfs):
                flow = df.vratiKonacniFlow()
                if self.gather is not None:
                    self.gather(flow, j)
            done = torch.cuda.Event()
            done.record()
            self.set_free[k] = done

    def sync(self):
        self.torch.cuda.synchronize()


class StubEngine:
    """CPU stand-in used by tests/test_bench_launcher.py: the same launcher, rendezvous, barrier / max-over-ranks timing,
    gather and JSON assembly as the GPU path, with gloo and a constant field instead of the HIP pipeline.  Never used for
    a measurement (the line it prints says data = "stub")."""

    def __init__(self, args, rank, local_rank, world):
        import torch
        self.torch = torch
        self.backend = "gloo"
        self.rank = rank
        self.P = 1
        self.field = torch.full((8, 8, 2), float(rank), dtype=torch.float32)
        self.gather = None
        self.jobs = None

    def make_jobs(self, passes):
        return [None] * len(passes)

    def consistency(self, fwd, bwd):
        return fwd - bwd

    def like(self):
        return self.field

    def step(self, i, timed):
        if self.gather is not None:
            self.gather(self.field, 0)
        return self.field

    def begin(self, nsteps):
        pass

    def sync(self):
        pass


# ---------------------------------------------------------------------------------------------------- fixed batch
def run_fixed_batch(eng, npasses, world, rank, use_dist, sync_all):
    """BASELINE configs[3] (README.md:40 of the reference: forward and backward runs are independent): a FIXED batch of
    npasses = 8 pairs x (forward, backward) passes, pass p -> rank p mod world, every finished field gathered on rank 0
    (RCCL), the forward/backward consistency check of every pair there (postprocessing.py:123-135).  The total work does
    not depend on the number of GPUs: strong scaling, next to the headline's weak scaling.  Same engine as the headline
    (a rank's passes in equal groups, front ends on HIP streams, batched sweeps).  Returns the wall time (max over ranks)."""
    import torch.distributed as dist
    sharding = importlib.import_module(PKG + ".sharding")
    torch = eng.torch
    mine = sharding.assign_passes(npasses, world, rank)
    rounds = -(-npasses // world)
    saved_gather, saved_jobs = eng.gather, eng.jobs
    eng.jobs = eng.make_jobs(mine)
    bufs = [sharding.make_gather_buffers(eng.like(), world, rank) for _ in range(eng.P)] if use_dist else None
    fields, calls = {}, [0]

    def gather(flow, slot):
        k = calls[0]
        calls[0] += 1
        if not use_dist:
            fields[mine[k]] = flow.clone()
            return
        sharding.gather_flows(flow, bufs[slot], rank)
        if rank == 0:
            for src in range(world):
                if k * world + src < npasses:
                    fields[k * world + src] = bufs[slot][src].clone()

    eng.gather = gather

    def once():
        calls[0] = 0
        fields.clear()
        eng.begin(len(mine))
        for i in range(len(mine)):
            eng.step(i, False)
        eng.sync()
        for _ in range(len(mine), rounds):                 # ranks with a shorter share keep the gathers collective
            gather(torch.zeros_like(eng.like()), 0)
        out = None
        if rank == 0:
            out = [eng.consistency(fields[2 * q], fields[2 * q + 1]) for q in range(npasses // 2)]
        return out

    once()
    sync_all()
    t0 = time.perf_counter()
    sparse = once()
    sync_all()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.like().device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.gather, eng.jobs = saved_gather, saved_jobs
    rec = {"workload": "BASELINE configs[3]: fixed batch of %d passes = %d 1024x436 pairs x (forward, backward), bcd_times=%d, pass p -> rank "
                       "p mod %d, flow fields gathered on rank 0, forward/backward consistency check (threshold 10) of every pair there"
                       % (npasses, npasses // 2, BCD_TIMES, world),
           "passes": npasses, "passes_per_rank": len(mine), "n_gpus": world, "ms": dt * 1e3, "ms_per_pass": dt * 1e3 / npasses,
           "Mpix/s": npasses * H * W / dt / 1e6, "scaling": "strong"}
    if rank == 0 and sparse is not None and hasattr(sparse[0], "shape") and sparse[0].dim() == 3 and sparse[0].shape[-1] == 3:
        rec["consistent_fraction_mean"] = float(sum(float(s_[..., 2].mean()) for s_ in sparse) / len(sparse))
    return rec


# ---------------------------------------------------------------------------------------------------- worker
def worker(args):
    import torch
    import torch.distributed as dist
    sharding = importlib.import_module(PKG + ".sharding")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.stub and os.environ.get("DFLOW_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)                       # tests/test_bench_launcher.py: a failing rank must fail the launcher
    eng = (StubEngine if args.stub else (BatchEngine if args.mode == "batch" else GpuEngine))(args, rank, local_rank, world)
    # under a launcher (ours or torch.distributed.run) the process group is always created, also for one rank, which
    # exercises the same RCCL gather path; a plain `python bench.py` runs without torch.distributed
    use_dist = "RANK" in os.environ
    if use_dist:
        # the communication libraries print banners (RCCL: version / host, gloo: connection lines) on STDOUT when the first
        # communicator comes up; stdout must carry the JSON line only, so fd 1 points at stderr until that has happened
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if eng.backend == "nccl":
                dist.init_process_group("nccl", device_id=eng.dev)
            else:
                dist.init_process_group("gloo")
            dist.barrier()
            eng.sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        bufs = [sharding.make_gather_buffers(eng.like(), world, rank) for _ in range(eng.P)]
        eng.gather = lambda flow, slot: sharding.gather_flows(flow, bufs[slot], rank)

    eng.begin(args.warmup)
    for i in range(args.warmup):
        eng.step(i, False)

    def sync_all():
        eng.sync()
        if use_dist:
            dist.barrier()
            eng.sync()

    def timed_region(nsteps, record):
        """Exactly nsteps steps between barrier + synchronize on both sides; the maximum over the ranks."""
        sync_all()
        t0 = time.perf_counter()
        eng.begin(nsteps)
        for i in range(nsteps):
            eng.step(i, record)
        sync_all()
        d = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([d], dtype=torch.float64, device=eng.like().device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    dt = timed_region(args.steps, True)
    # the driver's 20 steps are 0.2 s: one more region of the default length (98 steps, 14 groups of 7: fill and drain of the
    # two-group pipeline weigh < 10 %) right behind it, reported beside the headline, never instead of it
    dt_long = timed_region(LONG_STEPS, False) if (args.steps != LONG_STEPS and not args.no_long_run and not args.stub) else None

    fixed = run_fixed_batch(eng, args.fixed_batch, world, rank, use_dist, sync_all) if args.fixed_batch > 0 else None

    if rank == 0:
        out = {
            "metric": "Mpix/s flow (1024x436, bcd_times=4)",
            "value": world * args.steps * H * W / dt / 1e6,
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "stub" if args.stub else ("synthetic (rehearsal: all ranks on one GPU)" if args.rehearse_on_one_gpu else "synthetic"),
        }
        if dt_long is not None:
            out["ms_per_step_%d" % LONG_STEPS] = dt_long / LONG_STEPS * 1e3
            out["value_%d_steps" % LONG_STEPS] = world * LONG_STEPS * H * W / dt_long / 1e6
        if fixed is not None:
            out["fixed_batch"] = fixed
        if not args.stub:
            finish_report(out, eng, args, world)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def knn_kernel_times(torch, df, pair, reps=3):
    """The kernels of the kNN stage one by one, HIP events between them on the launch stream (dflow_knn_proposals_timed),
    one pair alone on the GPU: ({kernel: mean ms}, MFMA instructions the screen issues)."""
    st = torch.cuda.Stream(device=df.device)
    acc, issued = {}, 0.0
    with torch.cuda.stream(st):
        df.load_pair(*pair)
        df.generisi_timed()                               # warm-up
        for _ in range(reps):
            ms, issued = df.generisi_timed()
            for k, v in ms.items():
                acc[k] = acc.get(k, 0.0) + v / reps
    st.synchronize()
    return acc, issued


def profile_scalar(name, default=None):
    """A number from the tracked rocprofv3 PMC summary of this round (profiles/pmc_knn.json), e.g. the clock the chip held."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_knn.json")) as f:
            return json.load(f).get(name, default)
    except Exception:
        return default


def other_configs(eng, args):
    """One-GPU timings of the BASELINE configurations the headline does not cover, so that they are driver-observed:
    configs[2] (forward + backward pass of one 1024x436 pair + consistency check, postprocessing.py:123-135) and the
    geometry of configs[4] (1242x375, cells 54x25, fp16 descriptors, bcd_times=8).  Wall clock around synchronised runs."""
    torch, pl, synth = eng.torch, eng.pipeline, eng.synth
    _lib = importlib.import_module(PKG + "._lib")
    dev = eng.dev
    res = {}

    def run_passes(dfs, jobs, bcd_times, streams):
        evs = []
        for j, (df, (a, b)) in enumerate(zip(dfs, jobs)):
            st = streams[j % len(streams)]
            with torch.cuda.stream(st):
                df.load_pair(a, b); df.generisi(); df.nasumicni(); df.pakovanje()
                e = torch.cuda.Event(); e.record(); evs.append(e)
        main = torch.cuda.current_stream(dev)
        for e in evs:
            main.wait_event(e)
        pl.ceoBCD_batch(dfs, bcd_times)
        return [df.vratiKonacniFlow() for df in dfs]

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), sum(ts) / len(ts)

    streams = eng.front[:2] if hasattr(eng, "front") else [torch.cuda.Stream(device=dev) for _ in range(2)]
    # configs[2]: both directions of the bench's first pair, then the consistency check
    dfs = (eng.sets[0][:2] if hasattr(eng, "sets") and len(eng.sets[0]) >= 2 else
           [pl.DiscreteFlow(H, W, eng.cellh, eng.cellw, device=dev, seed=0) for _ in range(2)])
    a, b = eng.pairs[0]

    def c2():
        fwd, bwd = run_passes(dfs, [(a, b), (b, a)], BCD_TIMES, streams)
        return pl.fb_consistency(fwd, bwd, 10.0, dfs[0].p)
    best, mean = timed(c2)
    res["configs[2]"] = {"workload": "forward + backward pass of one 1024x436 pair (bcd_times=%d) + forward/backward consistency check "
                                     "(threshold 10, README.md:65), 1 GPU: front ends on 2 HIP streams, sweeps of both passes in one "
                                     "batched launch per phase" % BCD_TIMES,
                         "ms": best, "ms_mean": mean, "passes": 2, "Mpix/s": 2 * H * W / best / 1e3,
                         "dtype": "f32 descriptors / f32 distance / f64 DP"}
    # low texture: the same configuration on a frame with saturated sky, a flat road band, blur and a repeated pattern
    # (synth.py "low_texture"; real KITTI frames have these, daisy i flann.py:26-27,66): one pair alone on the GPU, next to the
    # dense-texture pair measured the same way, and what the kNN screen did
    lt1, lt2, _ = synth.make_pair(H, W, seed=eng.seeds[0], style="low_texture")
    lt = (torch.from_numpy(lt1).to(dev), torch.from_numpy(lt2).to(dev))
    one = dfs[:1]
    dense_best, _ = timed(lambda: run_passes(one, [(a, b)], BCD_TIMES, streams[:1]))
    lt_best, lt_mean = timed(lambda: run_passes(one, [lt], BCD_TIMES, streams[:1]))
    lt_knn, _ = knn_kernel_times(torch, one[0], lt)
    with torch.cuda.stream(streams[0]):
        one[0].load_pair(*lt); one[0].generisi()
        st = one[0].knn_stats()
    res["low_texture"] = {"workload": "single 1024x436 pair, forward, bcd_times=%d, synth style low_texture (31 %% saturated = all-zero DAISY, 24 %% "
                                      "road band of +-2 grey levels, blurred box, 16-px repeated pattern), one pair alone on the GPU" % BCD_TIMES,
                          "ms_per_pass": lt_best, "ms_per_pass_mean": lt_mean, "dense_texture_ms_per_pass": dense_best,
                          "ratio_to_dense_texture": lt_best / dense_best,
                          "knn_kernels_ms": {k: round(v, 4) for k, v in lt_knn.items()},
                          "events_per_query_cell": st["events_per_query_cell"], "lists": st["lists"],
                          "lists_to_exact_search": st["lists_exact"], "whole_pass_fallback": bool(st["flags"] & 1),
                          "max_entries_per_lane": st["max_entries_per_lane"], "list_capacity_per_lane": st["list_capacity"],
                          "all_zero_queries": st["zero_queries"], "all_zero_candidates_removed": st["zero_candidates_removed"],
                          "query_cell_pairs_with_a_wave_of_their_own": st["heavy_pairs"],
                          "note": "round 3's screen sent this frame's whole pass to the brute-force kernel (knn_fix_kernel 81 ms, "
                                  "profiles/r04_lowtex_round3_code.jsonl)"}
    # configs[4] geometry: 4 passes (2 pairs, both directions)
    kh, kw, kch, kcw, ksweeps = 375, 1242, 25, 54, 8
    kdfs = [pl.DiscreteFlow(kh, kw, kch, kcw, device=dev, seed=0, flags=_lib.FLAG_DESCR_F16) for _ in range(4)]
    jobs = []
    for pr in range(2):
        i1, i2, _ = synth.make_pair(kh, kw, seed=synth.pair_seed(pr, 0))
        t1, t2 = torch.from_numpy(i1).to(dev), torch.from_numpy(i2).to(dev)
        jobs += [(t1, t2), (t2, t1)]
    best, mean = timed(lambda: run_passes(kdfs, jobs, ksweeps, streams), reps=2)
    res["configs[4] geometry"] = {"workload": "1242x375, cells 54x25 (discrete_flow.py:22-23,30-31), fp16 DAISY descriptors (DFLOW_FLAG_DESCR_F16), "
                                              "MFMA-screened exact kNN, bcd_times=8; 2 synthetic pairs x (forward, backward) = 4 passes on 1 GPU "
                                              "(on 8 GPUs: one pass per rank and step)",
                                  "ms": best, "ms_mean": mean, "passes": 4, "ms_per_pass": best / 4, "Mpix/s": 4 * kh * kw / best / 1e3,
                                  "dtype": "f16 descriptors / f32 distance / f64 DP"}
    del kdfs
    # 1920x1080: not a BASELINE configuration; the largest frame the workspace layout is sized for, 2 passes
    fh, fw, fch, fcw = 1080, 1920, 40, 30
    fdfs = [pl.DiscreteFlow(fh, fw, fch, fcw, device=dev, seed=0) for _ in range(2)]
    jobs = []
    for pr in range(2):
        i1, i2, _ = synth.make_pair(fh, fw, seed=synth.pair_seed(pr, 0))
        jobs.append((torch.from_numpy(i1).to(dev), torch.from_numpy(i2).to(dev)))
    best, mean = timed(lambda: run_passes(fdfs, jobs, BCD_TIMES, streams), reps=2)
    res["1920x1080"] = {"workload": "1920x1080 (not in BASELINE.json), cells 30x40 px = the bench's 64x27 grid, forward passes of 2 synthetic pairs, "
                                    "bcd_times=%d, 1 GPU; workspace %.1f GB per pass" % (BCD_TIMES, fdfs[0].ws_bytes / 1e9),
                        "ms": best, "ms_mean": mean, "passes": 2, "ms_per_pass": best / 2, "Mpix/s": 2 * fh * fw / best / 1e3,
                        "dtype": "f32 descriptors / f32 distance / f64 DP"}
    del fdfs
    return res


def finish_report(out, eng, args, world):
    torch = eng.torch
    P = eng.P
    out["config"] = {"workload": "single 1024x436 Sintel-shape pair per step per GPU, forward only, bcd_times=4 "
                                 "(BASELINE.json configs[1]); cells 64x27, 150 labels/px",
                     "pairs_in_flight_per_gpu": P, "pair_seeds_rank0": eng.seeds, "mode": args.mode,
                     "parallelism": ("one pass per step, steps in equal groups of at most %d: front end of a group's pairs (DAISY, kNN, neighbour proposals) on %d HIP streams, "
                                     "their compat lists on a fourth, the BCD sweeps of the group on a fifth: one batched launch per phase (chains x passes), two groups "
                                     "alternate so that front end and sweeps of consecutive groups overlap; flow fields gathered "
                                     "on rank 0" % (P, args.front)) if args.mode == "batch" else
                                    ("one pass per step; %d independent steps in flight per GPU on separate HIP streams; "
                                     "flow fields gathered on rank 0" % P)}
    # ---- bcd_chain_kernel: 4 sweeps x 4 phases = 16 launches between the two events of a group
    launches = 4 * BCD_TIMES
    # (start, end, passes in the launch): a batched launch carries the chains of several passes
    bcd_ms = sum(a.elapsed_time(b) for a, b, _ in eng.bcd_events) / max(1, len(eng.bcd_events)) / launches
    passes_per_launch = sum(n for _, _, n in eng.bcd_events) / max(1, len(eng.bcd_events))
    # algorithmic bytes of one phase launch: every pixel of half the image lines is visited once and
    # needs its labels: L*4 B flows + L*4 B costs + 16 B per pixel, L = 150  ->  1216 B per visited pixel
    alg_bytes = int(passes_per_launch * (H * W // 2) * (150 * 4 + 150 * 4 + 16))
    achieved = alg_bytes / (bcd_ms * 1e-3) / 1e9
    # the launches of the timed region: column phases (W+1)//2 resp. W//2 chains, row phases (H+1)//2 resp. H//2 chains,
    # 192 threads per chain, times the passes of a group
    traffic, traffic_src = pmc_traffic("bcd_chain_kernel", [((W + 1) // 2) * 192, ((H + 1) // 2) * 192], passes_per_launch)
    chain = {"bound": "hbm", "kernel": "bcd_chain_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": traffic_src,
             "launch_ms": bcd_ms, "passes_per_launch": passes_per_launch, "algorithmic_bytes_per_launch": alg_bytes,
             "launch_ms_per_pass": bcd_ms / passes_per_launch, "ms_per_step": bcd_ms * launches / passes_per_launch,
             "note": "launch_ms = HIP-event time of the 16 chain launches of a group of passes / 16, measured on the "
                     "launch stream (%s mode, %d pairs per group / in flight); traffic = rocprofv3 PMC bytes of one "
                     "such launch (profiles/)" % (args.mode, P)}
    # ---- knn_screen_kernel: one launch per step, timed alone with HIP events on its launch stream
    knn_ms, mfma_issued = knn_kernel_times(torch, eng.flows[0], eng.pairs[0])
    screen_ms = knn_ms["knn_screen_kernel"]
    alg_flops = 2 * 68 * knn_pairs(H, W, eng.cellh, eng.cellw)
    issued_flops = mfma_issued * 32 * 32 * 16 * 2
    s_traffic, s_traffic_src = pmc_traffic("knn_screen_kernel")
    screen = {"bound": "mfma", "kernel": "knn_screen_kernel", "achieved": alg_flops / (screen_ms * 1e-3) / 1e12,
              "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": alg_flops / (screen_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
              "traffic": s_traffic, "traffic_source": s_traffic_src,
              "launch_ms": screen_ms, "ms_per_step": screen_ms,
              "algorithmic_flops_per_launch": alg_flops, "issued_flops_per_launch": issued_flops,
              "issued_over_algorithmic": issued_flops / alg_flops,
              "issued_tflops": issued_flops / (screen_ms * 1e-3) / 1e12,
              "effective_clock_ghz_from_profiles": profile_scalar("knn_screen_kernel_effective_clock_ghz"),
              "mfma_pipe_busy_frac_from_profiles": profile_scalar("knn_screen_kernel_mfma_busy_frac"),
              "note": "launch_ms = HIP events around the kernel on its launch stream (dflow_knn_proposals_timed), one pair alone on "
                      "the GPU, mean of 3; algorithmic flops = 2*68 per (query, candidate) pair of the +-2-cell windows (SURVEY 8(d)); "
                      "issued = v_mfma_f32_32x32x16_f16 count x 32768 (two passes over the 40 leading principal components "
                      "+ 8 bound slots, K = 48, padded tiles included)"}
    dominant = screen if screen["ms_per_step"] >= chain["ms_per_step"] else chain
    out["roofline"] = dict(dominant)
    out["roofline"]["dominant_by"] = "ms per step measured in this run: %s %.3f, %s %.3f" % (
        screen["kernel"], screen["ms_per_step"], chain["kernel"], chain["ms_per_step"])
    out["roofline"]["kernels"] = {"knn_screen_kernel": screen, "bcd_chain_kernel": chain}
    out["roofline"]["knn_kernels_ms"] = {k: round(v, 4) for k, v in knn_ms.items()}
    out["roofline"]["stages"] = stage_rooflines(torch, eng.flows[0], eng.pairs[0], eng.cellh, eng.cellw)
    tot, top = total_traffic_per_pass()
    alg = out["roofline"]["stages"]["end_to_end"]["bytes"]
    out["roofline"]["total_traffic_per_pass"] = {"hbm_bytes": tot, "algorithmic_bytes": alg, "ratio": (tot / alg) if tot else None,
                                                 "largest_kernels": top,
                                                 "source": "sum over all kernels of profiles/pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE), per pass"}
    stale, detail = profiles_state()
    out["stale_profiles"] = stale
    out["roofline"]["stale_profiles"] = stale
    out["roofline"]["profiles_collected_with"] = detail
    out["latency_ms_single_pair"] = out["roofline"]["stages"]["ms_total"]
    out["latency_note"] = ("ms_per_step is pipelined throughput (%d pairs per group, two groups alternating); one pair alone on the GPU, "
                           "stage after stage on one stream, takes latency_ms_single_pair" % P)
    # flow of the bench's first pair (what the EPE numbers refer to)
    df = eng.flows[0]
    gpu_flow = df.run(eng.pairs[0][0], eng.pairs[0][1], BCD_TIMES).cpu().numpy()
    _, _, gt = eng.synth.make_pair(H, W, seed=eng.seeds[0])
    m = window_mask(gt, eng.cellh, eng.cellw)
    out["epe"] = {"gpu": {"all_pixels": epe_stats(gpu_flow, gt), "gt_inside_image_and_search_window": epe_stats(gpu_flow, gt, m)}}
    out["epe_delta_vs_oracle"] = None
    if world == 1 and not args.no_other_configs:
        out["other_configs"] = other_configs(eng, args)
        if out.get("fixed_batch"):
            # on 8 GPUs every rank has 2 of the 16 passes = the configs[2] case (one pair, both directions, on one GPU)
            c3 = dict(out["fixed_batch"])
            c3["projected_speedup_1_to_8_gpus"] = c3["ms"] / out["other_configs"]["configs[2]"]["ms"]
            c3["projection_note"] = ("this run's %d-pass time / configs[2]'s time (2 passes on one GPU, what every rank of 8 has); the "
                                     "measured curve is fixed_batch.ms of the runs with --gpus 1, 2, 4, 8" % c3["passes"])
            out["other_configs"]["configs[3]"] = c3
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"], out["epe"], out["epe_delta_vs_oracle"] = cpu_baseline(eng.synth, gpu_flow, eng.seeds[0], eng.cellh, eng.cellw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=LONG_STEPS)       # 14 groups of 7: the fill and drain of the two-group pipeline weigh < 10 %
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-run", action="store_true", help="skip the extra timed region of %d steps (ms_per_step_%d)" % (LONG_STEPS, LONG_STEPS))
    ap.add_argument("--no-other-configs", action="store_true", help="skip the one-GPU timings of BASELINE configs[2] and [4]")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent pairs in flight per GPU (each on its own HIP stream and workspace)")
    ap.add_argument("--mode", choices=("streams", "batch"), default="batch",
                    help="streams: --inflight independent pipelines; batch: groups of --batch pairs share the BCD launches")
    ap.add_argument("--batch", type=int, default=7,
                    help="largest group in --mode batch (the steps are split into equal groups of at most this many pairs)")
    ap.add_argument("--front", type=int, default=3, help="HIP streams for the front end of a group in --mode batch")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks on device 0, gather over gloo (a multi-rank rehearsal on a one-GPU box; not a measurement)")
    ap.add_argument("--fixed-batch", type=int, default=16,
                    help="passes of the fixed batch run after the timed steps (BASELINE configs[3]: 8 pairs x 2 directions, pass p -> rank "
                         "p mod N: strong scaling, reported under fixed_batch); 0: skip")
    ap.add_argument("--groups", default=None, help=argparse.SUPPRESS)    # experiment: explicit group sizes of the timed region
    ap.add_argument("--no-lists-stream", action="store_true", help=argparse.SUPPRESS)   # experiment: compat lists on the BCD stream itself
    ap.add_argument("--lists-on-front", action="store_true", help=argparse.SUPPRESS)    # experiment: compat lists on the front-end streams (before round 3: the default)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # nothing has touched the GPU in this process
    worker(args)


if __name__ == "__main__":
    main()
