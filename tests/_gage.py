"""GAGE-shaped synthetic workload for the merge-block (L1) path: one random genome of bacterial size, two assemblies
of it with a realistic spread of contig lengths (log-normal, a few hundred bases to a few hundred kb), different
contig boundaries, gaps, strands, per-assembly errors (0.3 - 3 % divergence between the two), runs of N, and -- for
every pair of contigs that overlap on the genome -- the merge block gam-merge's graph code would hand to
PctgBuilder::alignMergeBlock: the contig ids, the tail flags and the list of blocks (read-supported frames on both
contigs).  Merge blocks are grouped into graphs (maximal chains of overlapping contigs, in genome order = the order of
gam-merge's graphs_list) each holding one merge list (path).

BASELINE configs 1-4 (GAGE S. aureus / R. sphaeroides / human chr14) need data that cannot be fetched here; this is
their stand-in: same sizes and shapes, known answer.  numpy only (a 2.9 Mb genome builds in about a second).
"""
import numpy as np

COMP = np.array([1, 0, 3, 2, 4], dtype=np.uint8)  # A<->T, C<->G, N
LETTERS = np.frombuffer(b"ATCGN", dtype=np.uint8)


def to_ascii(codes):
    return LETTERS[codes].tobytes()


def _copy_with_errors(rng, genome, g0, g1, sub, indel):
    """genome[g0:g1] with substitutions and single-base indels; pos[k] = index in the copy of genome base g0+k (or of
    the next surviving base)."""
    seg = genome[g0:g1].copy()
    n = len(seg)
    m = rng.random(n) < sub
    seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    keep = rng.random(n) >= indel
    ins = rng.random(n) < indel
    counts = keep.astype(np.int64) + ins.astype(np.int64)
    out = np.repeat(seg, counts)
    start = np.cumsum(counts) - counts
    # where a base was kept AND followed by an insertion, the second copy is the inserted (random) base;
    # where it was deleted but an insertion follows, the single copy is the inserted base
    ins_at = start[ins] + keep[ins].astype(np.int64)
    out[ins_at] = rng.integers(0, 4, size=len(ins_at))
    return out, start


def _assembly(rng, genome, first, median, sigma, min_len, max_len, gap_lo, gap_hi, flip_prob, div, n_frac):
    """div(): divergence of the next contig from the genome (substitutions 70 %, single-base indels 15 % + 15 %)"""
    ctgs, at, G = [], first, len(genome)
    while at + min_len <= G:
        n = int(np.clip(rng.lognormal(np.log(median), sigma), min_len, max_len))
        n = min(n, G - at)
        d = div()
        seq, pos = _copy_with_errors(rng, genome, at, at + n, 0.7 * d, 0.15 * d)
        if rng.random() < n_frac and len(seq) > 2000:          # scaffold-gap style runs of N
            for _ in range(int(rng.integers(1, 4))):
                L = int(rng.integers(10, 400))
                s = int(rng.integers(500, len(seq) - 500 - L)) if len(seq) > 1000 + L else 0
                seq[s:s + L] = 4
        flipped = bool(rng.random() < flip_prob)
        if flipped:
            seq = COMP[seq][::-1].copy()
        ctgs.append(dict(g0=at, g1=at + n, seq=seq, flipped=flipped, pos=pos, n=len(seq)))
        at += n + int(rng.integers(gap_lo, gap_hi + 1))
    return ctgs


def to_contig(c, g):
    """contig coordinate of genome position g (inside [g0, g1))"""
    p = min(int(c["pos"][g - c["g0"]]), c["n"] - 1)
    return c["n"] - 1 - p if c["flipped"] else p


def problem(seed, genome_len=2_900_000, block_every=3000, min_overlap=700):
    """dict(genome, master, slave, graphs) -- graphs: list of graphs, each a list of merge lists, each a list of merge
    blocks dict(m_id, s_id, blocks=[(m_begin, m_end, s_begin, s_end, m_strand, s_strand, n_reads)], tails)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    # master: longer, cleaner contigs (Allpaths-LG like); slave: shorter (MSR-CA / SOAPdenovo like), each contig 0.3 - 3 %
    # away from the genome (log-uniform), 3 % of them ~7 % away (collapsed repeat / paralogue: homology < 95 -> rejected)
    master = _assembly(rng, genome, int(rng.integers(0, 300)), 22000, 1.1, 600, 320000, 50, 900, 0.45,
                       lambda: 0.001, 0.10)
    slave = _assembly(rng, genome, int(rng.integers(300, 2500)), 11000, 1.2, 500, 200000, 50, 1200, 0.45,
                      lambda: 0.07 if rng.random() < 0.03 else float(np.exp(rng.uniform(np.log(0.003), np.log(0.03)))), 0.15)
    pairs = []
    si0 = 0
    for mi, m in enumerate(master):
        while si0 < len(slave) and slave[si0]["g1"] <= m["g0"]:
            si0 += 1
        si = si0
        while si < len(slave) and slave[si]["g0"] < m["g1"]:
            s = slave[si]
            lo, hi = max(m["g0"], s["g0"]), min(m["g1"], s["g1"])
            if hi - lo >= min_overlap:
                pairs.append((lo, mi, si, hi))
            si += 1
    pairs.sort()
    graphs, cur, last = [], [], None
    for lo, mi, si, hi in pairs:
        m, s = master[mi], slave[si]
        span = hi - lo
        nb = int(np.clip(span // block_every + 1, 1, 12))
        # read coverage reaches nearly to the ends of the overlap (a merge block whose shorter unaligned tail is
        # 100..199 bases always fails in the reference: PctgBuilder.cc:781 vs :1404-1408) -- except for ~10 % of the
        # pairs, where it stops a few hundred bases short and the tails get their own alignments
        short = rng.random() < 0.10 and span > 3000
        e0 = lo + (int(rng.integers(250, 600)) if short else int(rng.integers(5, 60)))
        e1 = hi - (int(rng.integers(250, 600)) if short and rng.random() < 0.5 else int(rng.integers(5, 60)))
        inner = np.sort(rng.choice(np.arange(e0 + 90, e1 - 90), size=2 * nb - 2, replace=False)) if nb > 1 and e1 - e0 > 400 + 2 * nb else np.array([], dtype=np.int64)
        cuts = [e0] + [int(x) for x in inner] + [e1]
        blocks = []
        for k in range(0, len(cuts) - 1, 2):
            a, b = cuts[k], cuts[k + 1]
            if b - a < 60:
                continue
            mc = sorted((to_contig(m, a), to_contig(m, b)))
            sc = sorted((to_contig(s, a), to_contig(s, b)))
            same = m["flipped"] == s["flipped"]
            # a few blocks vote for the wrong orientation (mis-mapped mates): exercises the retry of findBestAlignment
            wrong = rng.random() < 0.03
            blocks.append((mc[0], mc[1], sc[0], sc[1], "+", "+" if (same != wrong) else "-", int(rng.integers(5, 80))))
        if not blocks:
            continue
        blocks.sort(key=lambda b: b[0])   # the graph lists a vertex's blocks in master-contig order
        mb = dict(m_id=mi, s_id=si, blocks=blocks, tails=(1, 1, 1, 1))
        if last is not None and (mi == last[0] or si == last[1]):
            cur.append(mb)
        else:
            if cur:
                graphs.append([cur])
            cur = [mb]
        last = (mi, si)
    if cur:
        graphs.append([cur])
    return dict(genome=genome, master=master, slave=slave, graphs=graphs)


def merge_blocks(pb):
    """all merge blocks of all graphs in graphs_list order, plus (graph, list) index of each"""
    flat, where = [], []
    for gi, g in enumerate(pb["graphs"]):
        for li, l in enumerate(g):
            for mb in l:
                flat.append(mb)
                where.append((gi, li))
    return flat, where
