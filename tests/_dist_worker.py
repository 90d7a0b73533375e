"""Worker for tests/test_multirank_cpu.py: run under torch.distributed.run with the gloo backend."""
import json
import os
import sys

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gam_ngs_amd import shard  # noqa: E402

if __name__ == "__main__":
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    first, last = shard.contiguous_shard(rank, world, 10)
    costs = [(7 * i * i + 3) % 101 + 1 for i in range(37)]
    mine = shard.lpt_partition(costs, world)[rank]
    dist.barrier()
    # pretend rank r took (1 + r) seconds for its share
    dt, cells, failed = shard.reduce_step_stats(1.0 + rank, float(sum(costs[i] for i in mine)), float(rank))
    out = dict(rank=rank, world=world, first=first, last=last, mine=mine, dt=dt, cells=cells, failed=failed, total=sum(costs))
    with open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
