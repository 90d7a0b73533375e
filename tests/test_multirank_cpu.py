"""The N>1 path on CPU: the library's partitioner (host only) and two gloo ranks that shard a task list with it, with
no data-path collective, and reduce the benchmark scalars exactly as bench.py does on the GPUs (max of time, sum of
cells).  The multi-context calls themselves (gamdp_multi_*) need GPUs: tests/test_gpu_multi.py."""
import json
import os
import subprocess
import sys

from gam_ngs_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))


def py_lpt(costs, parts):
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * parts
    out = [0] * len(costs)
    for i in order:
        r = min(range(parts), key=lambda k: (loads[k], k))
        out[i] = r
        loads[r] += costs[i]
    return out


def test_lpt_partition_is_balanced_deterministic_and_as_specified():
    costs = [51_250_000] * 13 + [1_000_000 * (i % 7 + 1) for i in range(50)] + [5]
    for parts in (1, 2, 3, 8):
        part = api.partition_lpt(costs, parts)
        assert part == py_lpt(costs, parts)           # decreasing weight, ties by index, least-loaded part, ties low
        assert part == api.partition_lpt(costs, parts)
        loads = [sum(c for c, p in zip(costs, part) if p == r) for r in range(parts)]
        assert max(loads) <= min(loads) + max(costs)  # LPT bound
    assert api.partition_lpt([], 4) == []


def test_equal_weights_split_evenly():
    # the benchmark's fixed pair list: equal cells -> shares differ by at most one pair, round-robin by index
    for n in (0, 1, 7, 8, 100):
        for parts in (1, 2, 3, 8):
            part = api.partition_lpt([5] * n, parts)
            sizes = [part.count(r) for r in range(parts)]
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
            assert part == [i % parts for i in range(n)]


def test_two_gloo_ranks(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(HERE, "_dist_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    outs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert outs[0]["uniform"] == [0, 2, 4, 6, 8] and outs[1]["uniform"] == [1, 3, 5, 7, 9]
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(37))
    for o in outs:  # every rank sees the reduced values; rank 0 prints them in bench.py
        assert o["dt"] == 2.0 and o["cells"] == float(o["total"]) and o["failed"] == 1.0
