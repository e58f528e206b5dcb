"""Multi-GPU sharding of independent controllers: one process per GPU, contiguous blocks of the
batch, no collective inside the control loop; one gather of the logs (to one rank, or to all) and one
all-reduce of a few statistics at the end (RCCL over xGMI with backend "nccl", gloo on CPU for tests)."""
import os

import torch
import torch.distributed as dist

# Run the collectives even in a one-rank group (set by diagnostics that want the RCCL path exercised
# on a single GPU); normally a one-rank job skips them.
FORCE_COLLECTIVES = False


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process if absent)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def shard_range(total, rank, world):
    """Contiguous, balanced block [lo, hi) of `total` controllers owned by `rank`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _active(group):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def gather_logs(local, total=None, group=None, dst=None, sizes=None):
    """Gather per-controller logs along dim 0.  `local`: dict name -> tensor [b_local, ...].

    dst=None: all-gather, every rank gets dict name -> [total, ...].  dst=r: gather to rank r only (the
    other ranks get their own shard back) - one exchange of the shard instead of world-size copies.
    sizes: per-rank shard lengths when the caller knows them (shard_range); otherwise they are exchanged
    first.  Shards may differ in length; they are padded to the largest for the collective and trimmed."""
    if not _active(group):
        return dict(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    any_t = next(iter(local.values()))
    if dist.get_backend(group) != "nccl" and any_t.is_cuda:      # gloo: stage through host memory
        host = gather_logs({k: v.cpu() for k, v in local.items()}, total, group, dst, sizes)
        return {k: v.to(any_t.device) for k, v in host.items()}
    if sizes is None:
        n_local = torch.tensor([any_t.shape[0]], device=any_t.device, dtype=torch.int64)
        got = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(got, n_local, group=group)
        sizes = torch.cat(got).tolist()
    sizes = [int(s) for s in sizes]
    assert sizes[rank] == any_t.shape[0], "shard length does not match `sizes`"
    nmax = max(sizes)
    out = {}
    for name, t in local.items():
        pad = t
        if t.shape[0] < nmax:
            pad = torch.cat([t, t.new_zeros((nmax - t.shape[0],) + tuple(t.shape[1:]))], 0)
        pad = pad.contiguous()
        if dst is None:
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad, group=group)
        else:
            parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
            dist.gather(pad, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if parts is None:
            out[name] = t
            continue
        out[name] = torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
        if total is not None:
            assert out[name].shape[0] == total
    return out


def reduce_stats(values, op="sum", group=None):
    """All-reduce a small list of python floats; returns floats."""
    if not _active(group):
        return list(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=group)
    return t.tolist()
