#!/usr/bin/env python3
"""Do the host phases of the D contexts of a gamdp_multi call overlap?  (gamdp_hostpool.h)

    python3 tools/multi_host_overlap.py [--devices 0,0,0,0] [--pairs 12500] [--calls-per-pair 8]

Runs the driver-shaped batch of bench.py's `mixed150` record (tests/_mixed.py) once through ONE context and once through a
MultiContext over `--devices` (a device may be named several times: on a one-GPU box the kernels of the D contexts share the GPU,
the host phases are what is measured), in a child process with GAMDP_DIAG_TIMING=1, and lays the contexts' host phases
(prepare + plan/stage, results) side by side on the process's steady clock: sum of the phases, length of their union, wall time."""
import argparse, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LINE = re.compile(r"libgamdp align: (\d+) tasks, (\d+) launches: prepare ([\d.]+) ms, plan\+stage ([\d.]+) ms, upload\+kernels\+download ([\d.]+) ms \[ctx (\S+) began ([\d.]+)\]")
RES = re.compile(r"libgamdp align: results ([\d.]+) ms \[ctx (\S+) began ([\d.]+)\]")


def child(devices, n_pairs, cpp, steps):
    import _mixed
    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L
    seqs, calls = _mixed.mixed_batch(20261004, n_pairs, cpp)
    n = len(calls)
    tasks = (L.Task * n)()
    _mixed.fill_tasks(tasks, calls)
    out = (L.Result * n)()
    if len(devices) == 1:
        ctx = gam.Context(devices[0])
        sset = gam.SequenceSet(ctx, seqs, ascii=False)
        call = lambda: ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, n, out, None)
    else:
        ctx = gam.MultiContext(devices)
        sset = gam.MultiSequenceSet(ctx, seqs, ascii=False)
        call = lambda: ctx.lib.gamdp_multi_align_batch(ctx.handle, sset.handle, sset.handle, tasks, n, out)
    for k in range(steps + 1):
        sys.stderr.write("== step %d begins\n" % k); sys.stderr.flush()
        t0 = time.perf_counter()
        rc = call()
        dt = (time.perf_counter() - t0) * 1e3
        assert rc == 0, rc
        sys.stderr.write("== step %d wall %.3f ms\n" % (k, dt)); sys.stderr.flush()
    print(json.dumps([tuple(out[k].key()) for k in range(0, n, 97)]))


def union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + (cur_b - cur_a if cur_b is not None else 0.0)


def run(devices, a):
    env = dict(os.environ, GAMDP_DIAG_TIMING="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--devices", ",".join(map(str, devices)), "--pairs", str(a.pairs),
                        "--calls-per-pair", str(a.calls_per_pair), "--steps", str(a.steps)], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    steps, cur = [], None
    for l in r.stderr.splitlines():
        if l.startswith("== step") and "begins" in l:
            cur = dict(host=[], wall=None, gpu=[])
            continue
        m = re.match(r"== step \d+ wall ([\d.]+) ms", l)
        if m:
            cur["wall"] = float(m.group(1)); steps.append(cur); continue
        m = LINE.search(l)
        if m and cur is not None:
            prep, plan, gpu, began = float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(7))
            cur["host"].append((began, began + prep + plan)); cur["gpu"].append(gpu); continue
        m = RES.search(l)
        if m and cur is not None:
            cur["host"].append((float(m.group(3)), float(m.group(3)) + float(m.group(1))))
    steps = steps[1:]   # the first step warms up
    rec = dict(devices=devices, contexts_seen=len(steps[-1]["gpu"]),
               wall_ms=sum(s["wall"] for s in steps) / len(steps),
               host_sum_ms=sum(sum(b - x for x, b in s["host"]) for s in steps) / len(steps),
               host_union_ms=sum(union(s["host"]) for s in steps) / len(steps))
    return rec, json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,0,0,0")
    ap.add_argument("--pairs", type=int, default=12500)
    ap.add_argument("--calls-per-pair", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    devs = [int(x) for x in a.devices.split(",")]
    if a.child:
        child(devs, a.pairs, a.calls_per_pair, a.steps)
    else:
        one, k1 = run(devs[:1], a)
        many, kd = run(devs, a)
        assert k1 == kd, "results differ between one context and %d" % len(devs)
        print(json.dumps(dict(one_context=one, multi=many, same_results=True,
                              note="host = prepare + plan/stage + results of every context on the process's steady clock; "
                                   "host_sum / host_union = how many contexts' host phases ran at the same time on average")))
