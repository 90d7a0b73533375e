"""bench.py's N > 1 path, executed: two ranks under torch.distributed.run exactly as the driver starts them on an 8-GPU
node, except that both use GPU 0 and talk over gloo (BENCH_SHARE_GPU=1).  What the ranks stand in for is the reference's
worker pool (lib/src/pctg/ThreadedBuildPctg.cc:143-197: N workers pulling pairs, no exchange between them).

The NCCL branch differs from what runs here in the backend name, the device of the three reduced scalars and the
`device_ids` of the barrier -- nothing else (bench.py: `share_gpu`).  The numbers of these lines are NOT multi-GPU
measurements; the tests read the line's bookkeeping."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, ranks=2, timeout=900):
    """`python bench.py --gpus N ...` as a fresh CHILD process (it spawns torch.distributed.run before any GPU call)."""
    env = dict(os.environ, BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GAMDP_LIB", "GAMDP_QUAD_MIN"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--no-l1", "--no-mixed150", "--no-proxy", "--no-band150", "--arena-gb", "4"] + list(flags)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2500:]     # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


def check_common(line, ranks):
    assert line["metric"] == "GCUPS" and line["unit"] == "GCUPS" and line["higher_is_better"] is True
    assert line["n_gpus"] == ranks and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert line["config"]["failed_pairs"] == 0
    assert line["launch_info"] and all(r["kernel"].startswith("k_align") for r in line["launch_info"])
    assert line["roofline"]["kernel"].startswith("k_align") and line["roofline"]["launches"] >= 2
    assert 0 < line["roofline"]["frac"] < 1
    # the line is certified: a sample of rank 0's share went through the CPU path and agreed
    assert line["verified_pairs"] >= 16 and line["verified_against"] in ("reference", "port")
    assert "cpu_baseline" not in line            # a timing of the host cores: N = 1 only
    assert "BENCH_SHARE_GPU" in line["config"]["note"]


def test_two_ranks_strong_scaling_share_one_list():
    P, length = 192, 6000
    line = run_bench("--scaling", "strong", "--pairs", str(P), "--len", str(length), "--verify-pairs", "24")
    check_common(line, 2)
    assert line["scaling"] == "strong"
    assert line["config"]["pairs_total_per_step"] == P
    assert line["config"]["pairs_on_rank0_per_step"] == P // 2          # the LPT deal of equal weights: every second pair
    # value = the whole list's cells over the slowest rank's time: both ranks' cells are in the sum
    cells = line["config"]["cells_per_pair"] * P
    assert abs(line["value"] * 1e9 * line["ms_per_step"] / 1e3 - cells) / cells < 0.02
    # the strong line carries a short weak-scaling record: every rank its own P pairs
    w = line["weak"]
    assert w["pairs_per_gpu_per_step"] == P and w["value"] > 0 and w["steps"] == 1


def test_two_ranks_weak_scaling_own_pairs():
    P, length = 96, 6000
    line = run_bench("--scaling", "weak", "--pairs", str(P), "--len", str(length), "--verify-pairs", "16")
    check_common(line, 2)
    assert line["scaling"] == "weak" and "weak" not in line
    assert line["config"]["pairs_total_per_step"] == 2 * P
    assert line["config"]["pairs_on_rank0_per_step"] == P
    cells = line["config"]["cells_per_pair"] * 2 * P
    assert abs(line["value"] * 1e9 * line["ms_per_step"] / 1e3 - cells) / cells < 0.02


def test_three_ranks_uneven_share():
    """100 pairs over 3 ranks: rank 0 gets 34, the others 33 -- the deal, the sums and the verification hold."""
    line = run_bench("--scaling", "strong", "--pairs", "100", "--len", "5000", "--verify-pairs", "16", ranks=3)
    check_common(line, 3)
    assert line["config"]["pairs_on_rank0_per_step"] == 34


def test_the_nccl_calls_of_the_n_gt_1_branch_on_one_rank():
    """What the ranks above did over gloo, a real 8-GPU run does over RCCL: a float64 all_reduce (MAX, SUM) of three scalars on the rank's
    GPU and barriers that name the device.  Two ranks cannot share a GPU under RCCL, so the calls themselves run here on ONE rank
    (world_size 1, backend "nccl", a child process): the dtype, the ops and the `device_ids` barrier are accepted by this torch / RCCL
    build and give back what went in -- so that the only thing the first real SCALE run adds is more ranks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', init_method='tcp://127.0.0.1:%d', rank=0, world_size=1)\n"
        "dist.barrier(device_ids=[0]); torch.cuda.synchronize(); dist.barrier(device_ids=[0])\n"
        "t = torch.tensor([1.5, 51249940.0 * 100000, 0.0], dtype=torch.float64, device='cuda')\n"
        "a = t.clone(); dist.all_reduce(a, op=dist.ReduceOp.MAX)\n"
        "b = t.clone(); dist.all_reduce(b, op=dist.ReduceOp.SUM)\n"
        "assert a.tolist() == t.tolist() and b.tolist() == t.tolist(), (a, b)\n"
        "assert bench.reduce_step_stats(1.5, 7.0, 0.0, device='cuda') == (1.5, 7.0, 0.0)\n"
        "dist.barrier(device_ids=[0]); dist.destroy_process_group(); print('NCCL-OK')\n"
    ) % (ROOT, port)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "NCCL-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
