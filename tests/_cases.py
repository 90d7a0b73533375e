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
