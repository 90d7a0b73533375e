#!/usr/bin/env python3
"""Focused parity run for the direction-free / packed kernels of one band (default 512; `parity_band512.py 9 150` with
GAMDP_QUAD_MIN=1 for the four- and eight-task band-150 kernels): random windowed cases of 0.6-14 kb (so that the
direction-free range is empty, one group, a few groups ...; partners of a wavefront of very different length), with and
without N, with and without edit strings, and once more with a tiny scratch arena (few resident slots, several launches)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases  # noqa: E402
import _oracle as O  # noqa: E402
import _gpu  # noqa: E402
from _gpu import oracle_for, run_cases  # noqa: E402


def main():
    n_total = 0
    band = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        cases = _cases.window_cases(seed, band)
        if seed % 3 == 2:
            _gpu.ctx().set_arena_bytes(40 << 20)   # a handful of slots
        for want_ops in (False, True):
            res = run_cases(cases, want_ops=want_ops)
            for cs, r in zip(cases, res):
                o, ops = oracle_for(cs, want_ops)
                if o.status == O.INVALID:
                    continue
                n_total += 1
                if r.key() != o.key() or (want_ops and r.ops != ops):
                    print("MISMATCH seed", seed, {k: v for k, v in cs.items() if k not in ("a", "b")}, len(cs["a"]), len(cs["b"]), r.key(), o.key())
                    sys.exit(1)
    print("band-%d parity passed:" % band, n_total, "comparisons")


if __name__ == "__main__":
    main()
