"""shard.py — frame sharding across GPUs: stereo pairs are independent, so a
stream of pairs is dealt round-robin to ranks (pair i -> rank i mod N), every rank runs its shard
on its own GPU with its own sv_ctx, and results are returned in submission order.  There is no
collective on the data path; torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only to
collect the results / the timing on rank 0."""


def shard_indices(n_frames, world, rank):
    """Indices of the frames rank `rank` owns (pair i -> rank i mod world)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank %d/%d" % (world, rank))
    return list(range(rank, n_frames, world))


def merge_in_order(n_frames, world, shards):
    """Inverse of shard_indices: shards[r] is the list of results of rank r, in its own order."""
    if len(shards) != world:
        raise ValueError("need one shard per rank")
    out = [None] * n_frames
    for r in range(world):
        idx = shard_indices(n_frames, world, r)
        if len(shards[r]) != len(idx):
            raise ValueError("rank %d returned %d results for %d frames" % (r, len(shards[r]), len(idx)))
        for i, v in zip(idx, shards[r]):
            out[i] = v
    return out


def run_sharded(frames, compute_fn, dist=None):
    """Run compute_fn(left, right) over this rank's shard of `frames` (a list of (left, right));
    returns the full ordered result list on rank 0 (None elsewhere), or on every rank if
    dist is None (single process)."""
    if dist is None or not dist.is_initialized():
        return [compute_fn(l, r) for l, r in frames]
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = [compute_fn(*frames[i]) for i in shard_indices(len(frames), world, rank)]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    return merge_in_order(len(frames), world, gathered) if rank == 0 else None


def max_over_ranks(seconds, dist=None, device=None):
    """Timing rule of bench.py: the job takes as long as its slowest rank."""
    if dist is None or not dist.is_initialized():
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
