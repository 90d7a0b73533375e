"""Timing experiment (a library built with -DGAMDP_EXP_PHASES): where the wavefronts of the eight-task band-150 kernel
spend their time.   GAMDP_LIB=... python tools/phase_times.py [pairs] [len]"""
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
# the first task of every octet carries the record; octets are formed in the launch's sorted order, so find them by their fields
recs = [out[k] for k in range(P) if out[k].begin_a > 1000 and out[k].score > 0 and out[k].length < 10**7]
n = len(recs)
f = lambda g: sum(g(r) for r in recs) / max(n, 1) / 100.0
print("kernel %.1f ms; %d octet records of %d expected" % (ms, n, P // 8))
print("per octet (us): fill %.0f  end cells %.0f  walk_many %.0f (of which materialise %.0f, %d calls)  tails %.0f" % (
    f(lambda r: r.begin_a), f(lambda r: r.begin_b), f(lambda r: r.score), f(lambda r: r.first_a), sum(r.n_match for r in recs) / max(n, 1), f(lambda r: r.length)))
