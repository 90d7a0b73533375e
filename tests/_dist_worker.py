"""Worker for tests/test_multirank_cpu.py: run under torch.distributed.run with the gloo backend.

Does what a rank of `bench.py --scaling strong` does, minus the GPU: derives its share of a task list from the
library's own partitioner (gamdp_partition_lpt -- host only, deterministic, so the ranks agree without talking),
then reduces the timing scalars with the benchmark's reduction."""
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gam_ngs_amd import api  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    costs = [(7 * i * i + 3) % 101 + 1 for i in range(37)]
    part = api.partition_lpt(costs, world)
    mine = [i for i, p in enumerate(part) if p == rank]
    uniform = bench.rank_share(10, rank, world, [5] * 10)
    dist.barrier()
    # pretend rank r took (1 + r) seconds for its share
    dt, cells, failed = bench.reduce_step_stats(1.0 + rank, float(sum(costs[i] for i in mine)), float(rank), device="cpu")
    out = dict(rank=rank, world=world, uniform=uniform, mine=mine, dt=dt, cells=cells, failed=failed, total=sum(costs))
    with open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
