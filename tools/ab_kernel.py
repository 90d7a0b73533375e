#!/usr/bin/env python3
"""Kernel time of one batch shape under the library GAMDP_LIB names (A/B of two builds on ONE box: tools/ab_kernel.sh).
    GAMDP_LIB=... python3 tools/ab_kernel.py <pairs> <len> [band] [reps]      or      ... mixed [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
tag = os.path.basename(os.environ.get("GAMDP_LIB", "libgamdp.so"))
ctx = gam.Context(0)
if sys.argv[1] == "mixed":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _mixed
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    seqs, calls = _mixed.mixed_batch(20261004, 12500, 8)
    sset = gam.SequenceSet(ctx, seqs, ascii=False)
    P = len(calls)
    tasks = (L.Task * P)()
    _mixed.fill_tasks(tasks, calls)
    shape = "mixed150 %d calls" % P
else:
    P, length = int(sys.argv[1]), int(sys.argv[2])
    band = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
    tasks = (L.Task * P)()
    for k in range(P):
        t = tasks[k]; t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * k + 1] - 1
    shape = "%d x %d band %d" % (P, length, band)
out = (L.Result * P)()
ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)   # warm-up
ms0, n0 = ctx.kernel_time()
best, tot, wall, walls = 1e9, 0.0, 1e9, []
for rep in range(reps):
    t0 = time.perf_counter()
    assert ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None) == 0
    walls.append((time.perf_counter() - t0) * 1e3)
    wall = min(walls)
    ms1, n1 = ctx.kernel_time()
    best = min(best, ms1 - ms0); tot += ms1 - ms0; ms0 = ms1
chk = sum(out[k].score for k in range(0, P, max(1, P // 4096)))
names = "+".join("%s %.2f ms" % (r["kernel"], r["kernel_ms"]) for r in ctx.launch_info())   # (of the last call)
if os.environ.get("AB_WALLS"):
    print("walls:", " ".join("%.1f" % w for w in walls))
print("%-22s %-28s kernels: best %8.2f ms  mean %8.2f ms   call: best %8.2f median %8.2f mean %8.2f ms   (checksum %d)  %s" % (tag, shape, best, tot / reps, wall, sorted(walls)[len(walls) // 2], sum(walls) / len(walls), chk, names))
