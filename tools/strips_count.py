import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _mixed
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
ctx = gam.Context(0)
seqs, calls = _mixed.mixed_batch(20261004, 12500, 8)
sset = gam.SequenceSet(ctx, seqs, ascii=False)
P = len(calls); tasks = (L.Task * P)(); _mixed.fill_tasks(tasks, calls); out = (L.Result * P)()
for _ in range(2):
    assert ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None) == 0
print(os.path.basename(os.environ.get("GAMDP_LIB", "libgamdp.so")), [(r["kernel"], r["strips"], round(r["kernel_ms"], 2)) for r in ctx.launch_info()])
