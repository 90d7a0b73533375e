"""Seeded merge-list scenarios for the post-alignment stage (list surgery + buildPctgs + writers)."""
import random

FLAGS = ("align_rev", "align_ok", "m_ltail", "m_rtail", "s_ltail", "s_rtail")


def assemblies(rng, n_master=None, n_slave=None):
    nm = n_master or rng.randint(2, 6)
    ns = n_slave or rng.randint(2, 6)
    master = [[rng.choice((0, 1, 2, 3, 3, 2, 1, 0, 4)) for _ in range(rng.randint(60, 400))] for _ in range(nm)]
    slave = [[rng.choice((0, 1, 2, 3)) for _ in range(rng.randint(60, 400))] for _ in range(ns)]
    if rng.random() < 0.2:
        master.append([])  # an empty master contig never becomes a paired contig
    return master, slave


def block(rng, master, slave, m_id, s_id):
    ml, sl = len(master[m_id]), len(slave[s_id])
    a = rng.randrange(ml)
    b = rng.randrange(a, ml) if rng.random() < 0.95 else rng.randrange(0, a + 1)
    if rng.random() < 0.6:   # slave region of similar length (the 3 % rule decides who is asked)
        ln = max(1, int((b - a + 1) * rng.choice((1.0, 1.0, 0.98, 0.96, 1.02, 1.05, 0.9))))
        c = rng.randrange(sl)
        d = min(sl - 1, c + ln - 1)
    else:
        c = rng.randrange(sl)
        d = rng.randrange(c, sl)
    mb = dict(m_id=m_id, m_start=a, m_end=b, s_id=s_id, s_start=c, s_end=d, ext_slave_next=1, ext_slave_prev=1,
              m_rev=0, s_rev=0)
    mb["align_ok"] = int(rng.random() < 0.85)
    mb["align_rev"] = int(rng.random() < 0.3)
    for f in ("m_ltail", "m_rtail", "s_ltail", "s_rtail"):
        mb[f] = int(rng.random() < 0.6)
    return mb


def merge_list(rng, master, slave, n=None):
    """A path through the assembly graph: consecutive blocks share the master or the slave contig (mostly)."""
    nm = [i for i in range(len(master)) if master[i]]
    n = n or rng.randint(1, 8)
    m_id, s_id = rng.choice(nm), rng.randrange(len(slave))
    out = []
    for _ in range(n):
        out.append(block(rng, master, slave, m_id, s_id))
        r = rng.random()
        if r < 0.45:
            s_id = rng.randrange(len(slave))          # stay on the master contig
        elif r < 0.9:
            m_id = rng.choice(nm)                     # stay on the slave contig
        elif r < 0.95:
            pass                                      # same pair again
        else:
            m_id, s_id = rng.choice(nm), rng.randrange(len(slave))
    # make some lists monotone along the shared contig (what real paths look like), leave the rest wild
    if rng.random() < 0.5:
        for k in range(1, len(out)):
            p, q = out[k - 1], out[k]
            if p["m_id"] == q["m_id"] and q["m_start"] < p["m_start"]:
                for a, b in (("m_start", "m_start"), ("m_end", "m_end")):
                    p[a], q[b] = q[b], p[a]
    return out


def scenario(seed):
    rng = random.Random(seed)
    master, slave = assemblies(rng)
    graphs = [[merge_list(rng, master, slave) for _ in range(rng.randint(1, 4))] for _ in range(rng.randint(1, 3))]
    return master, slave, graphs


def vote(m_id, m_start, m_end, s_id, s_start, s_end):
    """Stands in for the host's read-pair evidence: any deterministic function of the arguments will do."""
    return (m_start + 3 * s_end + m_id + s_id) & 1


def vote_mb(mb):
    return vote(mb["m_id"], mb["m_start"], mb["m_end"], mb["s_id"], mb["s_start"], mb["s_end"])
