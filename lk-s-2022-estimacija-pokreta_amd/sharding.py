"""Multi-GPU driver: independent (pair, direction) passes are dealt round-robin to ranks, one process per GPU
(README.md:40 of the reference: forward and backward runs are independent and can be parallelised).  The only
communication is a gather of the finished (H,W,2) flow fields on rank 0 (RCCL over xGMI when the backend is
"nccl"; the same code runs on gloo for CPU tests).  No collective sits on the data path of a pass.
"""
import torch
import torch.distributed as dist


def assign_passes(npasses, world, rank):
    """Pass p runs on rank p mod world."""
    return list(range(rank, npasses, world))


def make_gather_buffers(like, world, rank):
    """Receive buffers for gather_flows (rank 0 only)."""
    if rank != 0:
        return None
    return [torch.empty_like(like) for _ in range(world)]


def gather_flows(flow, bufs, rank):
    """One flow field per rank -> rank 0."""
    dist.gather(flow, gather_list=bufs if rank == 0 else None, dst=0)
    return bufs


def run_passes(passes, compute, world, rank, like, compute_many=None):
    """Runs `compute(pass_descriptor) -> (H,W,2) tensor` for this rank's share of `passes` and gathers every
    result on rank 0.  Returns {pass index: tensor} on rank 0, {} elsewhere.  Ranks with fewer passes than
    the longest share send a dummy field in the last round(s) so that every gather is collective.
    `compute_many(list of descriptors) -> list of tensors`, if given, computes the rank's whole share at once (the GPU
    driver batches the BCD sweeps of a rank's passes into shared launches); results and gathers are the same."""
    mine = assign_passes(len(passes), world, rank)
    rounds = (len(passes) + world - 1) // world
    bufs = make_gather_buffers(like, world, rank) if world > 1 else None
    done = None
    if compute_many is not None:
        done = [f.clone() for f in compute_many([passes[i] for i in mine])]
    out = {}
    for r in range(rounds):
        idx = mine[r] if r < len(mine) else None
        if idx is None:
            flow = torch.zeros_like(like)
        elif done is not None:
            flow = done[r]
        else:
            flow = compute(passes[idx])
        if world == 1:
            out[idx] = flow.clone()
            continue
        gather_flows(flow.contiguous(), bufs, rank)
        if rank == 0:
            for src in range(world):
                p = r * world + src
                if p < len(passes):
                    out[p] = bufs[src].clone()
    return out
