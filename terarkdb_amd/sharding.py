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
    import torch
    import torch.distributed as dist

    fb = NO_BAD
    if first_bad_local is not None and 0 <= int(first_bad_local) < NO_BAD:
        fb = shard_lo + int(first_bad_local)
    mine = torch.tensor([int(mismatches), int(checked), fb], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        # one collective: every rank's three counters, summed / min-ed locally (a sum and a min
        # would otherwise need two all-reduces)
        world = dist.get_world_size()
        everyone = torch.empty(3 * world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(everyone, mine)
        rows = everyone.view(world, 3).cpu()
    else:
        rows = mine.view(1, 3).cpu()
    first = int(rows[:, 2].min().item())
    return VerifyTotals(int(rows[:, 0].sum().item()), int(rows[:, 1].sum().item()), -1 if first == NO_BAD else first)
