"""A batch shaped like the live driver's calls, for bench.py's `mixed150` record and tests/test_gpu_mixed_batch.py.

What gam-merge's merge-block driver asks of find_alignment (lib/src/pctg/PctgBuilder.cc):
  * chain calls (alignBlocks, :1652-1677): a window on the master that starts where the previous block's last match ended plus the
    gap between the blocks, the slave frame as the b window -- begin_a anywhere in the contig, a few hundred to a few thousand rows;
  * left tails (:1535-1568): force_end calls, the b window starts at base 0, begin_a = a findHits seed or the tail difference;
  * right tails (:1573-1611): force_start calls on a chop_begin() view of one contig (begin_a = 0 or a seed), the b window runs to
    the end of the other.
Here: contig pairs A / B with B a diverged copy of an inner stretch A[s:e] (so both tails exist), log-normal lengths, a few contigs with
runs of N, several calls per pair with random windows.  Deterministic in `seed`.  Bases are code bytes (A0 T1 C2 G3 N4).
"""
import numpy as np


def _mutate(rng, a, sub=0.02, ins=0.005, dele=0.005):
    b = a.copy()
    m = rng.random(len(b)) < sub
    b[m] = (b[m] + rng.integers(1, 4, int(m.sum()))) & 3
    b = b[rng.random(len(b)) >= dele]
    pos = np.nonzero(rng.random(len(b)) < ins)[0]
    if len(pos):
        b = np.insert(b, pos, rng.integers(0, 4, len(pos)).astype(np.uint8))
    return b


def mixed_batch(seed, n_pairs, calls_per_pair, band=150, len_lo=300, len_hi=20000, n_frac=0.01, force_frac=0.10):
    """-> (seqs, calls): seqs[2k] = A_k, seqs[2k+1] = B_k as bytes of codes; calls = list of dicts with the gamdp_task fields
    (a_id, b_id, a_off, begin_a, end_a, begin_b, end_b, fs, fe, band)."""
    rng = np.random.default_rng(seed)
    seqs, calls = [], []
    for k in range(n_pairs):
        la = int(np.clip(rng.lognormal(np.log(4000.0), 0.9), len_lo, len_hi))
        A = rng.integers(0, 4, la).astype(np.uint8)
        at_start = rng.random() < 0.3              # the overlap begins at base 0 of A (a merge block at the start of the master)
        s = 0 if at_start else int(rng.integers(0, max(1, la // 4)))
        e = la - int(rng.integers(0, max(1, la // 4)))
        B = _mutate(rng, A[s:e])
        if len(B) < 64:
            B = np.concatenate([B, rng.integers(0, 4, 64).astype(np.uint8)])
        if rng.random() < n_frac:                  # a scaffold: one run of N in either contig
            for S in (A, B):
                w = int(rng.integers(20, 400))
                p = int(rng.integers(0, max(1, len(S) - w)))
                S[p:p + w] = 4
        lb = len(B)
        seqs += [A.tobytes(), B.tobytes()]
        scale = (e - s) / float(lb)
        for _ in range(calls_per_pair):
            u = rng.random()
            c = dict(a_id=2 * k, b_id=2 * k + 1, a_off=0, fs=False, fe=False, band=band)
            if u < force_frac / 2:                 # left tail: force_end, b from base 0
                j1 = int(rng.integers(min(100, lb - 1), min(lb, 6000)))
                i1 = s + int(j1 * scale)
                seed_pos = int(rng.integers(0, 40))
                c.update(begin_a=max(0, i1 - j1 - seed_pos), end_a=max(0, i1 - 1), begin_b=0, end_b=j1 - 1, fe=True)
            elif u < force_frac:                   # right tail: force_start on a chopped view of A, b to the end of B
                j2 = int(rng.integers(max(0, lb - 6000), max(1, lb - 50)))
                i2 = min(la - 2, s + int(j2 * scale))
                tail = la - (i2 + 1)
                c.update(a_off=i2 + 1, begin_a=int(rng.integers(0, 30)) if rng.random() < 0.5 else 0, end_a=max(0, tail - 1),
                         begin_b=min(lb - 1, j2 + 1), end_b=lb - 1, fs=True)
            else:                                  # a chain call: a frame of B against the master window it maps to
                w = int(np.clip(rng.lognormal(np.log(3000.0), 0.8), 100, lb))
                x = 0 if rng.random() < 0.15 else int(rng.integers(0, max(1, lb - w + 1)))
                x = min(x, lb - 1)
                jitter = int(rng.integers(-20, 21))
                ba = max(0, s + int(x * scale) + jitter)
                c.update(begin_a=ba, end_a=min(la - 1, ba + int(w * scale) - 1), begin_b=x, end_b=min(lb - 1, x + w - 1))
            calls.append(c)
    return seqs, calls


def fill_tasks(tasks, calls):
    """calls -> the ctypes gamdp_task array `tasks` (gam_ngs_amd.lib.Task)"""
    for t, c in zip(tasks, calls):
        t.a_id, t.b_id, t.a_off, t.b_off = c["a_id"], c["b_id"], c["a_off"], 0
        t.a_rc = t.b_rc = 0
        t.force_start, t.force_end, t.band = int(c["fs"]), int(c["fe"]), c["band"]
        t.begin_a, t.end_a, t.begin_b, t.end_b = c["begin_a"], c["end_a"], c["begin_b"], c["end_b"]
