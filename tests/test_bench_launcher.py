"""`python bench.py --gpus N` must start its own ranks (the driver calls it that way) -- checked on CPU with the stub
engine: same launcher, env rendezvous (127.0.0.1), barrier + max-over-ranks timing, gather on rank 0 and JSON assembly as
the GPU path, over gloo with world_size 2."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                          timeout=300, env=e)


def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "Mpix/s" and d["value"] > 0 and d["data"] == "stub"
    assert abs(d["value"] - 2 * 3 * 436 * 1024 / (d["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * d["value"]
    # the fixed batch (BASELINE configs[3]): 16 passes whatever the number of ranks, pass p -> rank p mod 2: strong scaling
    fb = d["fixed_batch"]
    assert fb["passes"] == 16 and fb["passes_per_rank"] == 8 and fb["n_gpus"] == 2 and fb["scaling"] == "strong"
    assert abs(fb["Mpix/s"] - 16 * 436 * 1024 / (fb["ms"] * 1e-3) / 1e6) < 1e-6 * fb["Mpix/s"]


def test_fixed_batch_with_a_ragged_share():
    # 5 passes over 2 ranks: rank 0 has 3, rank 1 has 2 and must still take part in the third gather
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub", "--fixed-batch", "5"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    assert d["fixed_batch"]["passes"] == 5 and d["fixed_batch"]["passes_per_rank"] == 3


def test_bench_rank_failure_is_reported():
    # WORLD_SIZE/--gpus mismatch inside the children cannot happen through the launcher; a crashing rank must surface
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub"], env={"MASTER_PORT_OVERRIDE_TEST": "1", "DFLOW_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0


def test_bench_single_rank_under_external_launcher_env():
    # what torch.distributed.run gives a 1-rank job: RANK/WORLD_SIZE present -> the process-group path with one rank
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--stub"],
               env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout          # the communication library's banner must not reach stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2
