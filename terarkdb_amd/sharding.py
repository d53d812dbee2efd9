"""Block-range sharding of a verification corpus over the GPUs of one node.

Every SST block and every 32 KiB log block carries its own checksum, so a corpus splits
into contiguous ranges that are verified independently, one process per GPU.  No data
moves between GPUs; the only exchange is one all-gather of three counters per rank
(`torch.distributed`: RCCL over xGMI with the "nccl" backend, gloo on CPUs).
"""
from dataclasses import dataclass

NO_BAD = (1 << 63) - 1  # "no mismatch" in the signed 64-bit tensors the collectives carry


def shard_range(n_units, rank, world):
    """Contiguous [lo, hi) of rank's share of n_units; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


@dataclass
class VerifyTotals:
    mismatches: int
    checked: int
    first_bad: int  # global index of the first mismatching unit, -1 if none


def allreduce_totals(mismatches, checked, first_bad_local, shard_lo, device=None):
    """Combines per-rank verify counters.  first_bad_local is the index inside this rank's shard
    (or None / negative / the engine's ~0 when the shard is clean)."""
    import torch.distributed as dist

    fb = NO_BAD
    if first_bad_local is not None and 0 <= int(first_bad_local) < NO_BAD:
        fb = shard_lo + int(first_bad_local)
    if not (dist.is_available() and dist.is_initialized()):
        return VerifyTotals(int(mismatches), int(checked), -1 if fb == NO_BAD else fb)
    # one collective: every rank's three counters, summed / min-ed locally (a sum and a min would
    # otherwise need two all-reduces)
    world = dist.get_world_size()
    import torch
    mine = torch.tensor([int(mismatches), int(checked), fb], dtype=torch.int64, device=device)
    everyone = torch.empty(3 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(everyone, mine)
    rows = everyone.tolist()
    first = min(rows[2::3])
    return VerifyTotals(sum(rows[0::3]), sum(rows[1::3]), -1 if first == NO_BAD else first)


def gather_verify_counts(counts, shard_lo_of_rank):
    """Same result as allreduce_totals, straight from the engine's device counter block
    (int64 tensor {mismatches, first bad local index or ~0, checked}): one all-gather of the three
    words and one copy to the host.  shard_lo_of_rank(r) is the global index of rank r's first unit."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size()
        everyone = torch.empty(3 * world, dtype=torch.int64, device=counts.device)
        dist.all_gather_into_tensor(everyone, counts)
        rows = everyone.tolist()
    else:
        world = 1
        rows = counts.tolist()
    mism = sum(rows[0::3])
    checked = sum(rows[2::3])
    first = NO_BAD
    for r in range(world):
        fb = rows[3 * r + 1]
        if fb >= 0:  # the engine's "none" is ~0, i.e. -1 as a signed word
            first = min(first, shard_lo_of_rank(r) + fb)
    return VerifyTotals(mism, checked, -1 if first == NO_BAD else first)
