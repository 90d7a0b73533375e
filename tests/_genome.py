"""A small synthetic assembly-reconciliation problem with a known answer: one random genome, two assemblies of it
(different contig boundaries, gaps, strands and errors), and the merge list gam-merge's graph code would hand to
PctgBuilder for the chain of overlapping contigs.  Used by the end-to-end tests: alignment (L1) -> list surgery ->
buildPctgs -> .gam.fasta must give back a sequence that aligns to the genome over (nearly) its whole span."""
import random

COMP = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def copy_with_errors(rng, genome, g0, g1, sub=0.003, indel=0.0005):
    """genome[g0:g1] with substitutions / short indels; returns (sequence, pos) where pos[k] = index in the copy of
    genome base g0+k (or of the next surviving base)."""
    out, pos = [], []
    for k in range(g0, g1):
        pos.append(len(out))
        r = rng.random()
        if r < indel:
            continue                                    # deletion
        c = genome[k]
        if r < indel + sub:
            c = rng.choice([x for x in "ACGT" if x != c])
        out.append(c)
        if rng.random() < indel:
            out.append(rng.choice("ACGT"))              # insertion
    return "".join(out), pos


def assembly(rng, genome, first, min_len, max_len, min_gap, max_gap, flip_prob):
    """Contigs covering the genome left to right with gaps; each may be stored reverse-complemented."""
    ctgs, at = [], first
    while at + min_len <= len(genome):
        n = min(rng.randint(min_len, max_len), len(genome) - at)
        seq, pos = copy_with_errors(rng, genome, at, at + n)
        flipped = rng.random() < flip_prob
        ctgs.append(dict(g0=at, g1=at + n, seq=revcomp(seq) if flipped else seq, flipped=flipped, pos=pos, n=len(seq)))
        at += n + rng.randint(min_gap, max_gap)
    return ctgs


def to_contig(c, g):
    """contig coordinate of genome position g (inside [g0, g1))"""
    p = min(c["pos"][g - c["g0"]], c["n"] - 1)
    return c["n"] - 1 - p if c["flipped"] else p


def problem(seed, genome_len=30000):
    rng = random.Random(seed)
    genome = "".join(rng.choice("ACGT") for _ in range(genome_len))
    master = assembly(rng, genome, rng.randint(0, 300), 3000, 7000, 150, 500, 0.4)
    slave = assembly(rng, genome, rng.randint(800, 1500), 2500, 6000, 150, 500, 0.4)
    # one merge block (graph vertex) per pair of contigs that overlap enough, in genome order = the merge path
    pairs = []
    for mi, m in enumerate(master):
        for si, s in enumerate(slave):
            lo, hi = max(m["g0"], s["g0"]), min(m["g1"], s["g1"])
            if hi - lo >= 700:
                pairs.append((lo, mi, si, hi))
    pairs.sort()
    mbs = []
    for lo, mi, si, hi in pairs:
        m, s = master[mi], slave[si]
        nb = rng.randint(1, 3)
        # read coverage reaches nearly to the ends of the overlap (the reference rejects a merge block whose shorter
        # unaligned tail is 100..199 bases: too long to ignore, PctgBuilder.cc:781, too short to be aligned, :1404-1408)
        cuts = sorted(rng.sample(range(lo + 90, hi - 90), 2 * nb - 2) + [lo + rng.randint(5, 60), hi - rng.randint(5, 60)])
        blocks = []
        for k in range(0, 2 * nb, 2):
            a, b = cuts[k], cuts[k + 1]
            if b - a < 60:
                continue
            mc = sorted((to_contig(m, a), to_contig(m, b)))
            sc = sorted((to_contig(s, a), to_contig(s, b)))
            # reads map to both assemblies on the same strand iff the two contigs have the same orientation
            blocks.append((mc[0], mc[1], sc[0], sc[1], "+", "+" if m["flipped"] == s["flipped"] else "-", rng.randint(10, 60)))
        if not blocks:
            continue
        # the graph lists a vertex's blocks in master-contig order
        blocks.sort(key=lambda b: b[0])
        mbs.append(dict(m_id=mi, s_id=si, blocks=blocks, tails=(1, 1, 1, 1)))
    return dict(genome=genome, master=master, slave=slave, merge_list=mbs)


def piece_on_genome(ctg, start, end, src_rev):
    """Genome interval a paired-contig piece covers: the piece is [start, end] of the contig (reverse-complemented
    first when src_rev).  Returns (g_from, g_to, forward): walking the piece left to right moves along the genome from
    g_from to g_to."""
    import bisect
    n = ctg["n"]

    def genome_of(x):                       # x = coordinate in the copy the piece was cut from
        p = n - 1 - x if src_rev else x     # coordinate in the stored contig
        q = n - 1 - p if ctg["flipped"] else p
        return ctg["g0"] + min(bisect.bisect_left(ctg["pos"], q), len(ctg["pos"]) - 1)
    return genome_of(start), genome_of(end), (bool(src_rev) == bool(ctg["flipped"]))


def check_walk(pb, pctg, max_jump=40):
    """Every piece of a paired contig must continue where the previous one stopped, all in one direction along the
    genome.  Returns (first genome position, last genome position, forward, largest jump)."""
    walk = []
    for (cid, start, end, rev, is_master), src in zip(pctg.rows, pctg.src_rev):
        ctg = (pb["master"] if is_master else pb["slave"])[cid]
        walk.append(piece_on_genome(ctg, start, end, src))
    fwd = walk[0][2]
    assert all(w[2] == fwd for w in walk), walk
    worst = 0
    for a, b in zip(walk, walk[1:]):
        jump = (b[0] - a[1] - 1) if fwd else (a[1] - b[0] - 1)
        worst = max(worst, abs(jump))
        assert abs(jump) <= max_jump, (jump, walk)
    return walk[0][0], walk[-1][1], fwd, worst
