"""Loader for the committed golden vectors (tests/golden/*.json, generated from the reference)."""
import ctypes
import json
import os
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

KEYS = ("status", "begin_a", "begin_b", "score", "n_match", "length", "first_a", "first_b", "first_found",
        "last_a", "last_b", "last_found", "homology")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def expect_key(e):
    return tuple(e[k] for k in KEYS)


def check_ops(e, ops):
    if "ops" in e:
        assert ops == e["ops"]
    else:
        assert zlib.crc32(ops.encode()) == e["ops_crc32"]


def l0_cases():
    """All small L0 cases (hand-built + random) as (name, case-dict-with-bytes, expect)."""
    out = []
    for fn in ("l0_handbuilt.json", "l0_random.json"):
        for d in load(fn):
            c = dict(a=d["a"].encode(), b=d["b"].encode(), band=d["band"], begin_a=d["begin_a"], end_a=d["end_a"],
                     begin_b=d["begin_b"], end_b=d["end_b"], fs=d["fs"], fe=d["fe"])
            out.append((d["name"], c, d["expect"]))
    return out


def adversarial_cases():
    """(spec dict, case-dict-with-bytes, expect) of tests/golden/l0_adversarial.json: the inputs are rebuilt from the recipe
    (tests/_cases.py adversarial_pair) and checked against the CRC32 the generator recorded."""
    import _cases
    out = []
    for d in load("l0_adversarial.json"):
        c = _cases.adversarial_case(d["kind"], d["n"], d["band"])
        assert (len(c["a"]), len(c["b"]), zlib.crc32(c["a"]), zlib.crc32(c["b"])) == (d["a_len"], d["b_len"], d["a_crc32"], d["b_crc32"]), d["kind"]
        out.append((d, c, d["expect"]))
    return out
