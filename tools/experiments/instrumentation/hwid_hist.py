"""Placement experiment (library built with `make variant NAME=hwid FLAGS=-DGAMDP_EXP_HWID`): how many tasks every SIMD ran in
a short band-512 launch, and when they finished.   GAMDP_LIB=gam_ngs_amd/libgamdp_diag_hwid.so python tools/hwid_hist.py [pairs]"""
import collections, os, sys
sys.path.insert(0, os.getcwd())
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
P = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
length, band = 50000, 512
ctx = gam.Context(0)
sset = gam.SequenceSet.synthetic(ctx, 0, P, length)
tasks = (L.Task * P)()
for k in range(P):
    t = tasks[k]; t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
    t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * k + 1] - 1
out = (L.Result * P)()
for rep in range(2):
    ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
ms, n = ctx.kernel_time()
per_simd = collections.Counter()
tmax = collections.defaultdict(int)
t0 = min(out[k].first_b for k in range(P))
for k in range(P):
    hw, xcc = out[k].last_b & 0xffffffff, out[k].last_a & 0xf
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    key = (xcc, se, sh, cu, simd)
    per_simd[key] += 1
    tmax[key] = max(tmax[key], (out[k].first_b - t0) & 0x7fffffff)
hist = collections.Counter(per_simd.values())
print("kernel %.1f ms per launch (two launches); %d SIMDs seen" % (ms / max(1, n), len(per_simd)))
print("tasks per SIMD -> SIMDs:", sorted(hist.items()))
byn = collections.defaultdict(list)
for key, c in per_simd.items():
    byn[c].append(tmax[key] / 100.0)
for c in sorted(byn):
    v = sorted(byn[c]); print("  %2d tasks: last finish (us) min %.0f median %.0f max %.0f" % (c, v[0], v[len(v) // 2], v[-1]))

# time line of the pairs of a few SIMDs (task A's record: start, end of fill; both records: finish)
pairs = collections.defaultdict(list)
for k in range(P):
    if out[k].score == 12345:
        hw, xcc = out[k].last_b & 0xffffffff, out[k].last_a & 0xf
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
        pairs[key].append((out[k].begin_a, out[k].begin_b, out[k].first_b, hw & 15))
tz = min(v[0] for vs in pairs.values() for v in vs)
shown = collections.Counter()
for key in sorted(pairs):
    n = len(pairs[key])
    if shown[n] >= 3:
        continue
    shown[n] += 1
    print("SIMD", key, "%d pairs:" % n, "  ".join("[w%d start %.2f fill_end %.2f done(A) %.2f]" % (w, ((a - tz) & 0x7fffffff) / 1e5, ((b - tz) & 0x7fffffff) / 1e5, ((c - tz) & 0x7fffffff) / 1e5)
                                                for a, b, c, w in sorted(pairs[key])))
