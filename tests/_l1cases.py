"""Seeded synthetic merge blocks (master/slave contig pairs with block lists) for the L1 driver tests.

A scenario mimics what gam-merge's graph code hands to PctgBuilder::alignMergeBlock: two contigs that
overlap (optionally with the slave reverse-complemented), a list of blocks = pairs of frames in the
overlap, strands and read counts that vote for an orientation, and tail flags."""
import random

import _cases


def revcomp_str(s):
    comp = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}
    return "".join(comp[c] for c in reversed(s))


def scenario(rng, kind=None):
    """Returns dict(master=str, slave=str, blocks=[(m_begin,m_end,s_begin,s_end,m_strand,s_strand,n_reads)],
    tails=(m_l,m_r,s_l,s_r))."""
    kind = kind or rng.choice(["overlap", "overlap", "overlap_rev", "contained", "bad", "wrong_vote", "tiny", "reverse_order", "edge", "edge_bad"])
    core_len = rng.randint(400, 2500)
    core = _cases.rand_seq(rng, core_len, 0.01 if rng.random() < 0.2 else 0.0)
    div = rng.choice([0.2, 0.4, 0.6, 1.0])
    core_s = _cases.mutate(rng, core, 0.03 * div, 0.01 * div, 0.01 * div)
    ml, mr = rng.randint(0, 900), rng.randint(0, 900)
    sl, sr = rng.randint(0, 900), rng.randint(0, 900)
    if kind == "contained":
        sl, sr = rng.randint(0, 20), rng.randint(0, 20)
    if kind == "tiny":
        ml = mr = sl = sr = rng.randint(0, 3)
    if kind in ("edge", "edge_bad"):  # frames that run past the end of the master: can make the reference throw
        mr = rng.randint(0, 5)

    # tails: either unrelated or homologous (so the tail alignments succeed sometimes)
    def tail_pair(n_m, n_s):
        if rng.random() < 0.5:
            base = _cases.rand_seq(rng, max(n_m, n_s))
            sm = _cases.mutate(rng, base, 0.02, 0.005, 0.005)
            return base[len(base) - n_m:], (sm[-n_s:] if n_s else "")
        return _cases.rand_seq(rng, n_m), _cases.rand_seq(rng, n_s)
    mL, sL = tail_pair(ml, sl)
    mR, sR = tail_pair(mr, sr)
    mR, sR = mR[::-1], sR[::-1]
    master = mL + core + mR
    slave = sL + core_s + sR
    if kind in ("bad", "edge_bad"):
        slave = _cases.rand_seq(rng, len(slave))
    # blocks: frames inside the cores (approximately corresponding coordinates)
    nb = rng.randint(1, 4)
    cuts = sorted(rng.sample(range(50, core_len - 50), 2 * nb)) if core_len > 100 + 2 * nb else [10, core_len - 10]
    blocks = []
    scale = len(core_s) / float(core_len)
    for k in range(0, len(cuts) - 1, 2):
        mb, me = len(mL) + cuts[k], len(mL) + cuts[k + 1]
        sb, se = len(sL) + int(cuts[k] * scale), len(sL) + int(cuts[k + 1] * scale)
        blocks.append([mb, me, sb, min(se, len(slave) - 1), "+", "+", rng.randint(5, 50)])
    if kind in ("edge", "edge_bad"):
        blocks[-1][1] += rng.randint(50, 600)
        if rng.random() < 0.5:
            blocks[-1][0] = min(blocks[-1][0] + rng.randint(0, 400), len(master) - 1)
    slave_is_rev = kind == "overlap_rev" or (kind in ("contained", "reverse_order") and rng.random() < 0.5)
    if slave_is_rev:
        slave = revcomp_str(slave)
        n = len(slave)
        for b in blocks:
            b[2], b[3] = n - 1 - b[3], n - 1 - b[2]
            b[5] = "-"
    if kind == "wrong_vote":  # strands vote for the wrong orientation -> the retry path must fix it
        for b in blocks:
            b[5] = "-" if b[5] == "+" else "+"
    if kind == "reverse_order":
        blocks.reverse()
    tails = tuple(rng.random() < 0.6 for _ in range(4))
    return dict(kind=kind, master=master, slave=slave, blocks=[tuple(b) for b in blocks], tails=tails)


def scenarios(seed, n):
    rng = random.Random(seed)
    return [scenario(rng) for _ in range(n)]
