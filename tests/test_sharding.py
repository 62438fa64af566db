"""world_size-2 gloo tests of the multi-GPU driver (CPU): pass assignment and the flow-field gather.  The compute
function is the CPU oracle here (test infrastructure); on GPUs it is DiscreteFlow.run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def test_assign_passes():
    sh = pkg("sharding")
    assert sh.assign_passes(16, 8, 3) == [3, 11]
    assert sh.assign_passes(5, 2, 0) == [0, 2, 4] and sh.assign_passes(5, 2, 1) == [1, 3]
    allp = sorted(p for r in range(8) for p in sh.assign_passes(16, 8, r))
    assert allp == list(range(16))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, npasses, outdir, many=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from conftest import pkg as _pkg
    sh, synth = _pkg("sharding"), _pkg("synth")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, ch, cw = 36, 40, 9, 8
    p = O.make_params(H, W, ch, cw, seed=3)

    def compute(desc):
        pair, backward = desc
        a, b, _ = synth.make_pair(H, W, seed=synth.pair_seed(pair, 0), amp_x=4, amp_y=3)
        if backward:
            a, b = b, a
        return torch.from_numpy(O.full_pass(p, a, b, 1)["flows"][-1].astype(np.float32))

    passes = [(i // 2, i % 2) for i in range(npasses)]
    like = torch.empty((H, W, 2), dtype=torch.float32)
    if many:      # the rank's whole share at once, as the GPU driver does to batch the BCD sweeps of its passes
        got = sh.run_passes(passes, None, world, rank, like, compute_many=lambda ds: [compute(d) for d in ds])
    else:
        got = sh.run_passes(passes, compute, world, rank, like)
    if rank == 0:
        assert sorted(got) == list(range(npasses))
        for i, d in enumerate(passes):
            assert torch.equal(got[i], compute(d)), "gathered field %d differs" % i
        open(os.path.join(outdir, "ok"), "w").write("ok")
    else:
        assert got == {}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("npasses,many", [(4, False), (3, False), (5, True)])
def test_run_passes_gloo_world2(tmp_path, oracle, npasses, many):
    mp.spawn(_worker, args=(2, _free_port(), npasses, str(tmp_path), many), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
