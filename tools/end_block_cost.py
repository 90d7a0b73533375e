"""What the end blocks cost: the same pairs with end_a at the last base of a (the pos == end_a anti-diagonal crosses the band in the last
2 * band / 16 blocks, which run the capturing instance of the packed range) and with end_a far behind it (only the last row's blocks do).
   python tools/end_block_cost.py [band] [pairs] [len]"""
import sys, os
sys.path.insert(0, os.getcwd())
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
band = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 49152
length = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
ctx = gam.Context(0)
sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
tasks = (L.Task * P)()
out = (L.Result * P)()
for far in (0, 1, 0, 1):
    for k in range(P):
        t = tasks[k]; t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1 + (4 * band + 100000 if far else 0), 0, sset.lengths[2 * k + 1] - 1
    ms0, n0 = ctx.kernel_time()
    for rep in range(3):
        assert ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None) == 0
    ms1, n1 = ctx.kernel_time()
    print("band %d, %d pairs of %d: end_a %s: %.2f ms per call (kernels)" % (band, P, length, "far behind a" if far else "a's last base", (ms1 - ms0) / 3))
