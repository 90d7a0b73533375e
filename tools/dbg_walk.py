"""Debug helper (GPU box): where does the GPU's edit string first leave the oracle's, in traceback order?
python tools/dbg_walk.py [n] [band] [seed]"""
import os, random, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import _cases
from _gpu import run_cases, oracle_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
band = int(sys.argv[2]) if len(sys.argv) > 2 else 512
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = random.Random(seed)
if n == 0:   # the batch of test_medium_pairs_all_kernel_variants, band-512 members
    cases = []
    for nn, bb, nfrac in ((3000, 150, 0.0), (3000, 150, 0.01), (2500, 512, 0.0), (2500, 512, 0.02), (4000, 20, 0.0),
                           (3000, 64, 0.0), (2000, 100, 0.01), (3500, 250, 0.0), (1800, 400, 0.005), (6000, 543, 0.0)):
        a, b = _cases.related_pair(rng, nn, n_frac=nfrac)
        cases.append(dict(a=a.encode(), b=b.encode(), band=bb, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
        cases.append(dict(a=a.encode(), b=b.encode(), band=bb, begin_a=200, end_a=len(a) - 300, begin_b=190, end_b=len(b) - 310, fs=bool(nn % 2), fe=bool(bb % 2)))
    if os.environ.get("ONLY512"):
        cases = [c for c in cases if c["band"] == 512 and b"N" not in c["a"] and b"N" not in c["b"]]
    which = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    res = run_cases(cases, want_ops=True)
    cs, r = cases[which], res[which]
    a, b = cs["a"].decode(), cs["b"].decode()
    print("batch of", len(cases), "showing", which, "begin", cs["begin_a"], cs["begin_b"])
else:
    a, b = _cases.related_pair(rng, n)
    cs = dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False)
    cs2 = dict(cs)
    r = run_cases([cs, cs2], want_ops=True)[0]
o, ops = oracle_for(cs, True)
print("gpu", r.key()); print("ora", o.key())
C = 17 if band == 512 else 19
# walk the oracle's ops backwards from its end cell, tracking (x, y): ops in forward order; traceback = reversed
X = min(len(b), len(a) + band)
g, w = r.ops[::-1], ops[::-1]
# end cell of the oracle: begin + consumed
print("lens", len(r.ops), len(ops))
i = 0
while i < min(len(g), len(w)) and g[i] == w[i]:
    i += 1
# the oracle's end cell: (x, pos) at the end = begin + what the alignment consumes; ops over "ABMX": A = GAP_A (consumes b), B = GAP_B (consumes a)
na = sum(1 for ch in ops if ch in "BMX"); nb = sum(1 for ch in ops if ch in "AMX")
x, pos = o.begin_b + nb - 1 - cs['begin_b'], o.begin_a + na - 1     # row of b (begin_b = 0 here), position in a of the end cell
for ch in w[:i]:
    if ch in "MX": x -= 1; pos -= 1
    elif ch == "A": x -= 1
    else: pos -= 1
y = pos - x + band - cs['begin_a']   # band column: pos = begin_a + x + y - band, x = row relative to begin_b
l, c = y // C, y % C
tau = x + l
print("first divergence at traceback step", i, "of", len(w), ": gpu", g[i:i+16], "ora", w[i:i+16])
print("  cell x=%d pos=%d y=%d lane=%d col=%d tau=%d blk=%d group=%d t=%d strip(4)=%d lam=%d" % (x, pos, y, l, c, tau, tau >> 4, tau >> 6, tau & 63, l // 4, l % 4))
