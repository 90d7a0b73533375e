"""Timing experiment (a library built with -DGAMDP_EXP_PHASES, see README.md): where the wavefronts of the eight-task band-150
kernel spend their time, phase by phase.   GAMDP_LIB=... python3 tools/experiments/instrumentation/phase_times.py [pairs] [len]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
P = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
length = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
band = 150
ctx = gam.Context(0)
sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
tasks = (L.Task * P)()
for k in range(P):
    t = tasks[k]; t.a_id, t.b_id, t.band = 2*k, 2*k+1, band
    t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length-1, 0, sset.lengths[2*k+1]-1
out = (L.Result * P)()
for rep in range(2):
    ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
ms, n = ctx.kernel_time()
r0 = [out[k] for k in range(P) if out[k].last_b == 0x7e57]
r1 = [out[k] for k in range(P) if out[k].last_b == 0x7e58]
r2 = [out[k] for k in range(P) if out[k].last_b == 0x7e59]
n0, n1 = max(len(r0), 1), max(len(r1), 1)
f0 = lambda g: sum(g(r) for r in r0) / n0 / 100.0
f1 = lambda g: sum(g(r) for r in r1) / n1 / 100.0
print("kernels %.1f ms in %d launches; %d + %d unit records of %d units" % (ms, n, len(r0), len(r1), P // 8))
names = ["setup", "int32 blocks quad A", "int32 blocks quad B", "packed top blocks", "packed plain blocks", "packed end blocks", "int32 tail blocks",
         "tasks' values to LDS", "end cells", "walk_many", "strips (materialise)", "finish_walk x 8"]
vals = [f0(lambda r: r.begin_a), f0(lambda r: r.begin_b), f0(lambda r: r.score), f0(lambda r: r.n_match), f0(lambda r: r.length), f0(lambda r: r.first_a),
        f0(lambda r: r.first_b), f0(lambda r: r.last_a), f1(lambda r: r.begin_a), f1(lambda r: r.begin_b), f1(lambda r: r.score), f1(lambda r: r.n_match)]
tot = sum(vals)
for nm, v in zip(names, vals):
    print("  %-24s %9.1f us  %5.1f %%" % (nm, v, 100.0 * v / max(tot, 1e-9)))
print("  %-24s %9.1f us; %.1f strip calls per unit" % ("unit", tot, f1(lambda r: r.length) * 100.0))
f2 = lambda g: sum(g(r) for r in r2) / max(len(r2), 1) / 100.0
print("  of which: the plain range's blocks %.1f (the rest of that phase: its return), the end range's blocks %.1f; of the values' phase: before the loop %.1f, the loop %.1f"
      % (f2(lambda r: r.begin_a), f1(lambda r: r.last_a), f1(lambda r: r.first_a), f1(lambda r: r.first_b)))
