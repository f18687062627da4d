"""Multi-GPU plumbing: streams are independent (reference soniclib.c:61-82 / speedy.c:130-176 hold all state per
handle), so they are sharded across ranks with NO data-path collective.  torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU node, "gloo" in CPU tests) is used only for the work-partition handshake and for reducing
the totals / timing."""
import numpy as np


def partition(n_streams, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`: stream i -> GPU floor(i / ceil(n/world))."""
    per = -(-n_streams // world)
    lo = min(n_streams, rank * per)
    hi = min(n_streams, lo + per)
    return lo, hi


def partition_by_cost(costs, world):
    """Longest-processing-time-first assignment for ragged batches: returns a list of index lists per rank."""
    order = np.argsort(-np.asarray(costs, np.int64), kind="stable")
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += int(costs[i])
    return [sorted(x) for x in out]


def handshake(dist, n_local_streams, n_local_frames, device="cpu"):
    """all_gather of each rank's (stream count, input frames): every rank learns the global layout."""
    import torch
    t = torch.tensor([n_local_streams, n_local_frames], dtype=torch.int64, device=device)
    if dist is None or not dist.is_initialized():
        return t.cpu().numpy().reshape(1, 2)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def reduce_totals(dist, samples, seconds, device="cpu"):
    """SUM of samples processed, MAX of elapsed seconds over ranks."""
    import torch
    if dist is None or not dist.is_initialized():
        return int(samples), float(seconds)
    s = torch.tensor([samples], dtype=torch.int64, device=device)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s.item()), float(t.item())
