import sys, os
sys.path.insert(0, os.getcwd())
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
ctx = gam.Context(0)
P, length = 64, 50000
sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
tasks = (L.Task * P)()
for k in range(P):
    t = tasks[k]; t.a_id, t.b_id, t.band = 2*k, 2*k+1, 512
    t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length-1, 0, sset.lengths[2*k+1]-1
out = (L.Result * P)()
ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
print("materialise calls per task:", [out[k].n_match for k in range(16)], "length", out[0].length)
print("10-ns ticks in materialise:", [out[k].first_a for k in range(16)])
print("10-ns ticks in the whole walk:", [out[k].first_b for k in range(16)])
print("walk iterations:", [out[k].last_a for k in range(16)])
print("direction-cache refills:", [out[k].last_b for k in range(16)])
