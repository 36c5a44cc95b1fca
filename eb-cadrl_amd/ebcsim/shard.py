"""Multi-GPU = env-index slicing.  Scenes are independent (no cross-env term anywhere in
simulator/env.py:388-466), so rank r owns a contiguous slice of the env index and the
simulation path needs no collective.  torch.distributed is used only to agree on timing
(barrier, max over ranks) and to sum the processed units for the whole-job rate."""


def shard_range(n_total, rank, world):
    """Contiguous slice [start, start + count) of n_total envs owned by `rank`; the first
    n_total % world ranks get one extra env."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def weak_range(envs_per_rank, rank):
    """Weak scaling (bench.py): every rank brings its own envs_per_rank scenes."""
    return rank * envs_per_rank, envs_per_rank


def scene_seeds(base_seed, start, count):
    """Seed of global env e is base_seed + e, whichever rank owns it."""
    return [base_seed + start + i for i in range(count)]


def job_rate(elapsed_local, units_local, device=None):
    """(max elapsed over ranks, total units).  Falls back to the local values when
    torch.distributed is not initialised (single process)."""
    import torch
    import torch.distributed as dist
    import os
    forced = os.environ.get("EBCSIM_FORCE_COLLECTIVES") == "1"  # a one-rank group still goes through its backend
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not forced):
        return float(elapsed_local), float(units_local)
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
