"""Multi-GPU sharding of the path: merge blocks / contig pairs are independent
(lib/src/pctg/BuildPctgFunctions.cc:82-84 mutates only its own MergeBlock), so N GPUs = N processes,
each aligning a static share of the task list on its own device.  There is no data-path collective;
torch.distributed is used only for the benchmark's barrier and for reducing the timing scalars."""
from typing import List, Sequence


def contiguous_shard(rank: int, world: int, n: int):
    """[first, last) of rank's share of n equal-cost tasks (synthetic benchmark: weak scaling uses
    n = world * per_rank, so every rank gets exactly per_rank tasks)."""
    base, extra = divmod(n, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def lpt_partition(costs: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time partition of task indices by predicted cell count
    (x_size * y_size summed over a merge block's DP calls): deterministic, so every rank computes the
    same assignment without communicating."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return parts


def reduce_step_stats(dt: float, cells: float, failed: float, device=None):
    """(max over ranks of dt, sum of cells, sum of failed) -- identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt, cells, failed
    t = torch.tensor([dt, cells, failed], dtype=torch.float64, device=device)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone()
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return tmax[0].item(), tsum[1].item(), tsum[2].item()
