"""Diagnostics (libgamdp_diag.so + GAMDP_DIAG_COUNT_MAT=1): strip materialisations and walk cost per task.
    GAMDP_LIB=gam_ngs_amd/libgamdp_diag.so GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py [band] [pairs] [len]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
band = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
length = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
ctx = gam.Context(0)
sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
tasks = (L.Task * P)()
for k in range(P):
    t = tasks[k]; t.a_id, t.b_id, t.band = 2*k, 2*k+1, band
    t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length-1, 0, sset.lengths[2*k+1]-1
out = (L.Result * P)()
ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
ms, n = ctx.kernel_time()
avg = lambda f: sum(f(out[k]) for k in range(P)) / P
calls, mt, wt, it, rf = avg(lambda r: r.n_match), avg(lambda r: r.first_a), avg(lambda r: r.first_b), avg(lambda r: r.last_a), avg(lambda r: r.last_b)
print("band %d, %d pairs of %d: kernel %.1f ms; per task: %.1f materialise calls, %.1f us in them (%.1f us per call), whole walk %.1f us, "
      "%.0f walk iterations, %.0f direction-cache refills" % (band, P, length, ms, calls, mt / 100.0, mt / 100.0 / max(calls, 1e-9), wt / 100.0, it, rf))
print("  (packed kernels: 'whole walk' = time inside walk_values incl. materialise_v; iterations / refills are walk_values' own)")
print("  first tasks: calls", [out[k].n_match for k in range(8)], "iterations", [out[k].last_a for k in range(8)])
