"""Multi-GPU sharding of independent controllers: one process per GPU, contiguous blocks of the
batch, no collective inside the control loop; one all-gather of the logs and one all-reduce of a
few statistics at the end (RCCL over xGMI with backend "nccl", gloo on CPU for tests)."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process if absent)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def shard_range(total, rank, world):
    """Contiguous, balanced block [lo, hi) of `total` controllers owned by `rank`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_logs(local, total=None, group=None):
    """All-gather per-controller logs along dim 0.  `local`: dict name -> tensor [b_local, ...].
    Shards may differ by one controller; they are padded to the largest shard for the collective
    and trimmed afterwards.  Returns dict name -> tensor [total, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(local)
    world = dist.get_world_size(group)
    any_t = next(iter(local.values()))
    if dist.get_backend(group) != "nccl" and any_t.is_cuda:      # gloo: stage through host memory
        host = gather_logs({k: v.cpu() for k, v in local.items()}, total, group)
        return {k: v.to(any_t.device) for k, v in host.items()}
    n_local = torch.tensor([any_t.shape[0]], device=any_t.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    out = {}
    for name, t in local.items():
        pad = t
        if t.shape[0] < nmax:
            pad = torch.cat([t, t.new_zeros((nmax - t.shape[0],) + tuple(t.shape[1:]))], 0)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        out[name] = torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
        if total is not None:
            assert out[name].shape[0] == total
    return out


def reduce_stats(values, op="sum", group=None):
    """All-reduce a small list of python floats; returns floats."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(values)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=group)
    return t.tolist()
