"""The N>1 path on CPU: two gloo ranks shard a task list with no data-path collective and reduce the
benchmark scalars exactly as bench.py does on the GPUs (max of time, sum of cells)."""
import json
import os
import subprocess
import sys

from gam_ngs_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def test_contiguous_shards_cover_exactly():
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            spans = [shard.contiguous_shard(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_lpt_partition_is_balanced_and_deterministic():
    costs = [51_250_000] * 13 + [1_000_000 * (i % 7 + 1) for i in range(50)] + [5]
    parts = shard.lpt_partition(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= min(loads) + max(costs)  # LPT bound
    assert parts == shard.lpt_partition(costs, 8)


def test_two_gloo_ranks(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(HERE, "_dist_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    outs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [(o["first"], o["last"]) for o in outs] == [(0, 5), (5, 10)]
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(37))
    for o in outs:  # every rank sees the reduced values; rank 0 prints them in bench.py
        assert o["dt"] == 2.0 and o["cells"] == float(o["total"]) and o["failed"] == 1.0
