"""Seeded generators of L0 alignment cases shared by the oracle and the GPU parity tests.

A case is a dict: a, b (ASCII bytes over ACGTN...), band, begin_a, end_a, begin_b, end_b, fs, fe.
"""
import random

ALPHA = "ACGT"


def rand_seq(rng, n, n_frac=0.0):
    s = [rng.choice(ALPHA) for _ in range(n)]
    if n_frac > 0:
        i = 0
        while i < n:
            if rng.random() < n_frac:
                run = rng.randint(1, 6)
                for k in range(i, min(n, i + run)):
                    s[k] = "N"
                i += run
            i += 1
    return "".join(s)


def mutate(rng, s, sub=0.03, ins=0.01, dele=0.01):
    out = []
    for ch in s:
        u = rng.random()
        if u < dele:
            pass
        elif u < dele + sub:
            out.append(rng.choice([c for c in ALPHA if c != ch]))
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append(rng.choice(ALPHA))
    return "".join(out)


def related_pair(rng, n, n_frac=0.0, div=1.0):
    a = rand_seq(rng, n, n_frac)
    b = mutate(rng, a, 0.03 * div, 0.01 * div, 0.01 * div)
    if not b:
        b = "A"
    return a, b


def random_case(rng, max_len=300, bands=(0, 1, 2, 5, 8, 20, 150), windowed=True):
    """Random (often adversarial) case in the spirit of SURVEY Appendix A's validation set."""
    n = rng.randint(12, max_len)
    kind = rng.random()
    n_frac = 0.05 if rng.random() < 0.4 else 0.0
    if kind < 0.6:
        a, b = related_pair(rng, n, n_frac, div=rng.choice([0.0, 1.0, 1.0, 3.0]))
        # shift one of them so the optimal path is not on the main diagonal
        if rng.random() < 0.5:
            k = rng.randint(0, 12)
            if rng.random() < 0.5:
                a = rand_seq(rng, k) + a
            else:
                b = rand_seq(rng, k) + b
    elif kind < 0.8:
        a, b = rand_seq(rng, n, n_frac), rand_seq(rng, rng.randint(12, max_len), n_frac)
    else:
        # low-complexity / tie-rich
        a = "".join(rng.choice("AC") for _ in range(n))
        b = "".join(rng.choice("AC") for _ in range(rng.randint(12, max_len)))
    band = rng.choice(bands)
    la, lb = len(a), len(b)
    if windowed and rng.random() < 0.7:
        begin_a = rng.randint(0, la - 1)
        end_a = rng.randint(begin_a, la + 5) if rng.random() < 0.8 else rng.randint(0, la + 5)
        begin_b = rng.randint(0, lb - 1)
        end_b = rng.randint(begin_b, lb + 5) if rng.random() < 0.9 else rng.randint(0, lb + 5)
    else:
        begin_a, end_a, begin_b, end_b = 0, la - 1, 0, lb - 1
    fs = rng.random() < 0.25
    fe = rng.random() < 0.25
    return dict(a=a.encode(), b=b.encode(), band=band, begin_a=begin_a, end_a=end_a, begin_b=begin_b,
                end_b=end_b, fs=fs, fe=fe)


def cases(seed, n, **kw):
    rng = random.Random(seed)
    return [random_case(rng, **kw) for _ in range(n)]


def beyond_cases(seed, n, bands=(0, 1, 5, 20, 150), max_len=120):
    """Windows whose begin_a lies at or past the end of a: begin_a in [|a|, |a| + band + rows + 8] and far past it
    (the reference's `|a| + band - begin_a` wraps there, banded_smith_waterman.cc:93-95: the b window alone bounds the
    rows and no cell is inside a).  The last sequence pair of a set gets these too, so a kernel that fetched there
    would run off the allocation."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        la, lb = rng.randint(1, max_len), rng.randint(1, max_len)
        a, b = rand_seq(rng, la, 0.03 if rng.random() < 0.3 else 0.0), rand_seq(rng, lb)
        band = rng.choice(bands)
        begin_b = rng.randint(0, lb - 1)
        end_b = rng.randint(begin_b, lb + 3)
        rows = min(end_b, lb - 1) - begin_b + 1
        kind = rng.random()
        if kind < 0.6:
            begin_a = rng.randint(la, la + band + rows + 8)
        elif kind < 0.8:
            begin_a = la + band + rng.randint(-2, 2)
        else:
            begin_a = la + rng.choice([10 ** 4, 10 ** 6, 2 ** 30])
        begin_a = max(0, begin_a)
        k = rng.random()
        if k < 0.3:
            end_a = begin_a + rng.randint(0, 2 * band + rows + 4)
        elif k < 0.6:
            end_a = rng.randint(0, begin_a)
        elif k < 0.8:
            end_a = max(0, begin_a - band + rng.randint(-3, 3))
        else:
            end_a = la - 1
        out.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=begin_a, end_a=end_a, begin_b=begin_b,
                        end_b=end_b, fs=rng.random() < 0.3, fe=rng.random() < 0.4))
    return out


# ---- adversarial long pairs: built to attack the exactness argument of the packed-f16 blocks -- the
# value ranges a lane and a block can span, ties everywhere, paths hugging the band edges.  Deterministic in (kind, n, band).
ADVERSARIAL_KINDS = ("homopolymer", "ac_shift1", "ac_vs_ag", "unrelated", "complement", "ins400_b", "ins400_a",
                     "ins_past_band_b", "ins_past_band_a", "div50", "tandem17", "tandem19")


def adversarial_pair(kind, n, band):
    """(a, b) ASCII strings of about n bases."""
    rng = random.Random(zlib_seed(kind, n, band))
    comp = {"A": "T", "T": "A", "C": "G", "G": "C"}
    if kind == "homopolymer":
        return "A" * n, "A" * (n - n // 50)
    if kind == "ac_shift1":
        return "AC" * (n // 2), ("AC" * (n // 2 + 1))[1:n - 3]
    if kind == "ac_vs_ag":
        return "AC" * (n // 2), "AG" * (n // 2)
    if kind == "unrelated":
        return rand_seq(rng, n), rand_seq(rng, n)
    if kind == "complement":   # every column of the main diagonal mismatches
        a = rand_seq(rng, n)
        return a, "".join(comp[c] for c in a)
    if kind in ("ins400_b", "ins400_a", "ins_past_band_b", "ins_past_band_a"):
        # one long indel early on: the path runs along (400) or is pinned against (band + 40) a band edge from there on
        a = rand_seq(rng, n)
        b = mutate(rng, a, 0.02, 0.002, 0.002)
        k = 400 if kind.startswith("ins400") else band + 40
        at = n // 10
        if kind.endswith("_b"):
            b = b[:at] + rand_seq(rng, k) + b[at:]
        else:
            a = a[:at] + rand_seq(rng, k) + a[at:]
        return a, b
    if kind == "div50":
        a = rand_seq(rng, n)
        return a, mutate(rng, a, 0.30, 0.10, 0.10)
    if kind in ("tandem17", "tandem19"):   # periods = the columns a lane owns at band 512 / in the band-150 throughput kernels
        p = 17 if kind == "tandem17" else 19
        unit = rand_seq(rng, p)
        a = (unit * (n // p + 1))[:n]
        b = mutate(rng, a, 0.01, 0.003, 0.003)
        return a, b[p // 2:]
    raise ValueError(kind)


def zlib_seed(kind, n, band):
    import zlib
    return zlib.crc32(("%s/%d/%d" % (kind, n, band)).encode())


def adversarial_specs():
    """(kind, n, band) of every committed adversarial vector: 3-20 kb at band 512, 6-20 kb at band 150."""
    out = []
    for i, kind in enumerate(ADVERSARIAL_KINDS):
        out.append((kind, (3000, 8000, 20000)[i % 3], 512))
        out.append((kind, (12000, 3500)[i % 2], 512))
        out.append((kind, (6000, 12000, 20000)[(i + 1) % 3], 150))
        out.append((kind, (9000, 6500)[i % 2], 150))
    return out


def adversarial_case(kind, n, band):
    a, b = adversarial_pair(kind, n, band)
    return dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False)


def window_cases(seed, band, count=60):
    """Random windowed cases of 0.6 - 14 kb for the direction-free / packed kernels of one band (tools/parity_band512.py and
    tests/test_gpu_l0_parity.py::test_window_cases_on_long_pairs): lengths such that the direction-free range is empty, one
    group, a few groups ...; partners of a wavefront of very different length; begin_a = 0 half the time (wavefronts that
    share it run their top blocks packed), anywhere else otherwise; end_a anywhere from begin_a to past the end of a (an
    early end_a puts the pos == end_a anti-diagonal into the top blocks); windows on b; force flags; N in a third."""
    rng = random.Random(5120 + seed)
    cases = []
    for _ in range(count):
        n = rng.choice([600, 900, 1300, 2000, 3000, 5000, 8000, 14000])
        a = rand_seq(rng, n, 0.002 if rng.random() < 0.3 else 0.0)
        b = mutate(rng, a[rng.randint(0, 200):], 0.03, rng.choice([0.0, 0.01, 0.03]), rng.choice([0.0, 0.01, 0.03]))
        if not b:
            b = "A"
        ba = rng.choice([0, 0, rng.randint(0, 700), rng.randint(0, n - 1)])
        ea = rng.choice([n - 1, n - 1, rng.randint(ba, n + 600)])
        bb = rng.choice([0, 0, rng.randint(0, min(300, len(b) - 1))])
        eb = rng.choice([len(b) - 1, len(b) - 1, rng.randint(bb, len(b) + 50)])
        cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=ba, end_a=ea, begin_b=bb, end_b=eb,
                          fs=rng.random() < 0.2, fe=rng.random() < 0.2))
    return cases


def displaced_path_cases(band, seed=0, n=5200):
    """Pairs whose alignment runs parallel to the main diagonal of the band, k columns off it, for k from one band edge to the
    other (b = a copy of a[k:], or k unrelated bases in front of a copy of a): a path in every strip of the direction-free
    kernels, the first and the last lane of a task included (strips that begin sshift lanes before a multiple of their width
    have lanes there that belong to the neighbouring task or to nobody).  A third of the pairs carries N."""
    rng = random.Random(7700 + 31 * band + seed)
    ks = sorted(set([0, 1, 2, band - 1, band - 2, band - 3, band - 9, band - 20, band - 39] + [rng.randint(3, band - 3) for _ in range(7)]))
    cases = []
    for k in ks:
        if k < 0 or k >= band:
            continue
        for sign in (1, -1):
            a = rand_seq(rng, n + k, 0.002 if rng.random() < 0.33 else 0.0)
            if sign > 0:
                b = mutate(rng, a[k:], 0.02, 0.002, 0.002)
            else:
                b = rand_seq(rng, k) + mutate(rng, a, 0.02, 0.002, 0.002)
            cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1,
                              fs=False, fe=False))
    return cases


def n_edge_cases(band):
    """(a, b, begin_a, end_a, begin_b, end_b) with ONE N placed around each edge of what the DP touches: a[begin_a - band ..
    begin_a + rows - 1 + band] and b[begin_b .. begin_b + rows - 1].  Two shapes: the alignment runs along the band's first column
    (b is a copy of a from begin_a - band + 1 on) or along its last (a copy from begin_a + band - 1 on), so that the first / last
    bases of the a range are ON the path; an N there scores 0 where the A in its place in the 2-bit plane would score +5 or -4."""
    rng = random.Random(8800 + band)
    out = []
    rows = 700
    for shape in ("low", "high"):
        begin_a = band + 50 if shape == "low" else 40
        shift = -band + 1 if shape == "low" else band - 1          # b[x] pairs with a[begin_a + shift + x]
        a0 = rand_seq(rng, begin_a + shift + rows + band + 700)
        b0 = a0[begin_a + shift: begin_a + shift + rows + 350]     # b is longer than the window on it
        begin_b, end_b = 0, rows - 1
        first_a, last_a = begin_a - band, begin_a + rows - 1 + band    # the a range the DP touches
        spots = [("a", first_a + d) for d in (-400, -321, -257, -66, -65, -64, -2, -1, 0, 1, 2, 3, 9)] + \
                [("a", last_a + d) for d in (-9, -3, -2, -1, 0, 1, 2, 63, 64, 65, 66, 257, 321, 400)] + \
                [("b", end_b + d) for d in (-2, -1, 0, 1, 2, 64, 65, 66, 257, 300)] + [("b", d) for d in (0, 1, 2)]
        for which, p in spots:
            a, b = list(a0), list(b0)
            tgt = a if which == "a" else b
            if p < 0 or p >= len(tgt):
                continue
            tgt[p] = "N"
            out.append(("".join(a).encode(), "".join(b).encode(), begin_a, len(a) - 1, begin_b, end_b, (shape, which, p - (first_a if which == "a" else 0))))
    return out
