"""Multi-GPU partitioning of the path: independent streams, one context each
("Different data streams MUST use different ctx instances", reference
README.markdown:376), so stream i simply lives on rank i mod world and the only
exchange step is one all-reduce of the per-rank match counters.
One process per GPU; the collective goes through torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests).
"""


def shard_streams(nstreams, rank, world):
    """Global stream indices owned by `rank`: round-robin i mod world."""
    return list(range(rank, nstreams, world))


def local_stream_count(nstreams, rank, world):
    return (nstreams - rank + world - 1) // world if nstreams > rank else 0


def allreduce_counts(counts, device=None):
    """Sum a small vector of int64 counters over all ranks (no-op without an
    initialised process group).  Returns a list of ints."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(counts), dtype=torch.int64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def allreduce_max(value, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_ints(value, device=None):
    """One int64 per rank, gathered on every rank (which device ordinal each rank
    drives: bench.py reports it so that a rehearsal with ranks sharing a GPU is
    told apart from a scaling run)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [int(x.item()) for x in out]
    return [int(value)]


def allreduce_counter_vector(counters, device=None):
    """The path's counter vector: uint64[1 + nregexes] =
    [streams with a match, matches of regex 0, matches of regex 1, ...], summed over
    all ranks with ONE all-reduce."""
    return allreduce_counts(counters, device)


def gather_stream_records(local_records, nstreams, rank, world, device=None):
    """Per-stream results {rc, count, ovector...} of ALL streams on every rank, in
    GLOBAL stream order (one all-gather of n_streams/G x record).

    `local_records` are this rank's records in the order of shard_streams(nstreams,
    rank, world); every record has the same number of int64 slots.  Ranks that own
    one stream fewer are padded for the collective (all_gather wants equal shapes)
    and the padding is dropped again."""
    import torch
    import torch.distributed as dist
    per = (nstreams + world - 1) // world                 # records of the fullest rank
    slots = len(local_records[0]) if local_records else 0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        s = torch.tensor([slots], dtype=torch.int64, device=device or "cpu")
        dist.all_reduce(s, op=dist.ReduceOp.MAX)          # a rank without streams learns the width
        slots = int(s.item())
    t = torch.zeros((per, max(slots, 1)), dtype=torch.int64, device=device or "cpu")
    if local_records:
        t[:len(local_records), :slots] = torch.tensor(local_records, dtype=torch.int64, device=t.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
    else:
        parts = [t]
    out = [None] * nstreams
    for r in range(world):
        rows = parts[r].tolist()
        for i, g in enumerate(shard_streams(nstreams, r, world)):
            out[g] = [int(x) for x in rows[i][:slots]]
    return out
