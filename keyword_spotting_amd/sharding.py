"""Utterance-level data parallelism: streams are independent (no cross-stream term anywhere on the
path), so a batch is partitioned across ranks with NO data-path collective.  torch.distributed
(RCCL over xGMI on the GPU box, gloo in CPU tests) is used only for the timing barrier and the
8-byte throughput reduction."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(total, rank, world):
    """Contiguous, balanced partition of `total` streams: rank r owns [lo, hi)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request total=%d rank=%d world=%d" % (total, rank, world))
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seed(seed, rank):
    """Independent synthetic data per shard (seed + rank)."""
    return seed + rank


def barrier(dist, device=None):
    if dist is not None and dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_throughput(dist, frames, seconds, device):
    """(total frames over all ranks, max seconds over ranks)."""
    import torch
    if dist is None or not dist.is_initialized():
        return int(frames), float(seconds)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    s = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    return int(round(f.item())), float(s.item())


def count_ranks(dist, device):
    """all-reduce(SUM) of 1: how many ranks actually took part in the report (the bench line's `ranks_seen`)."""
    import torch
    if dist is None or not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())
