""".blocks reader / writer behind the C ABI against the Python restatement (oracle/blocks_oracle.py; parity with the
reference is unpinned for these few lines, see there)."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import blocks_oracle as BO  # noqa: E402
from gam_ngs_amd import api  # noqa: E402


def rand_block(rng):
    b = {"n_reads": rng.choice([0, 1, 5, 9, 10, 11, 250, 10 ** 7])}
    for side in "ms":
        beg = rng.randint(0, 10 ** 6)
        b[side + "_ctg"] = rng.randint(0, 5000)
        b[side + "_strand"] = rng.choice("+-?")
        b[side + "_begin"] = beg
        b[side + "_end"] = beg + rng.randint(0, 10 ** 5)
        b[side + "_block_reads_len"] = rng.randint(0, 10 ** 12)
        b[side + "_reads_len"] = rng.randint(0, 10 ** 13)
    return b


def test_write_then_read_round_trip(tmp_path):
    rng = random.Random(3)
    blocks = [rand_block(rng) for _ in range(500)]
    p = tmp_path / "a.blocks"
    api.write_blocks(p, blocks)
    assert p.read_text() == BO.render(blocks)
    for mbs in (1, 10, 11):
        got = api.load_blocks(p, mbs)
        assert got == [b for b in blocks if b["n_reads"] >= mbs] == BO.load_blocks(p.read_text(), mbs)


def test_reader_on_awkward_files(tmp_path):
    good = "12\t0\t3\t+\t100\t900\t5000\t7000\t0\t8\t-\t50\t870\t4800\t6900"
    lines = [
        "# a comment", "", good, good.replace("\t", "   "), "  " + good,               # blanks instead of tabs, leading blanks
        good + "\ttrailing junk 1 2 3",                                               # extra fields are ignored
        "12\t0\t3\t+\t100\t900",                                                      # too short: dropped
        good.replace("\t+\t", "\t+"),                                                 # strand glued to the next number
        good.replace("100", "1e3"),                                                   # 1 then 'e' as the next token: shifts, fails
        good.replace("5000", "-1"),                                                   # unsigned field wraps
        good.replace("900", "99999999999"),                                           # int32 overflow: dropped
        "x" + good, "9\t" + good, "#" + good, "\t", "7",
        good.replace("12\t", "+12\t"), good.replace("12\t", "-3\t"),
    ]
    rng = random.Random(8)
    for _ in range(200):                                                              # mutated good lines
        t = list(good)
        for _ in range(rng.randint(1, 3)):
            k = rng.randrange(len(t))
            t[k] = rng.choice("0123456789+-?\t xe.#")
        lines.append("".join(t))
    text = "\n".join(lines)                                                           # no newline at the end of the file
    p = tmp_path / "awkward.blocks"
    p.write_text(text)
    for mbs in (-5, 1, 10, 13):
        want = BO.load_blocks(text, mbs)
        got = api.load_blocks(p, mbs)
        assert got == want, mbs
    assert len(BO.load_blocks(text, 1)) >= 8
    p2 = tmp_path / "crlf.blocks"
    p2.write_bytes((good + "\r\n" + good + "\r\n").encode())                          # '\r' is a blank for >>
    assert api.load_blocks(p2, 1) == BO.load_blocks(good + "\r\n" + good + "\r\n", 1)
    assert len(api.load_blocks(p2, 1)) == 2
