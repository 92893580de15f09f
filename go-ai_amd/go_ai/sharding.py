"""Multi-GPU plumbing: games are sharded by id, results gathered once (no collective in the search)."""
import torch

STAT_FIELDS = ("leaf_evals", "seconds", "games", "finished", "black_wins", "white_wins", "total_steps")


def shard_games(total_games, world_size, rank):
    """Game-id range [lo, hi) owned by `rank`: contiguous, sizes differ by at most one
    (the reference gives every rank ceil(episodes / world), /root/reference/go_ai/utils.py:190-197)."""
    base, extra = divmod(total_games, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_stats(record, device=None):
    """The run's single collective: all-gather one small record per rank (the counterpart of
    the four scalar allreduces at /root/reference/go_ai/utils.py:201-204).  Works on RCCL
    (backend 'nccl', device tensors) and on gloo (CPU tensors); without an initialised process
    group it returns the local record only.  Returns one dict per rank."""
    import torch.distributed as dist
    mine = torch.tensor([float(record.get(k, 0)) for k in STAT_FIELDS], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    return [dict(zip(STAT_FIELDS, p.cpu().tolist())) for p in parts]


def reduce_stats(per_rank):
    """Whole-job totals: throughput = all leaf evals / slowest rank's time."""
    total = {k: sum(r[k] for r in per_rank) for k in STAT_FIELDS if k != "seconds"}
    total["seconds"] = max(r["seconds"] for r in per_rank)
    total["leaf_evals_per_s"] = total["leaf_evals"] / total["seconds"] if total["seconds"] > 0 else 0.0
    return total
