"""Host-side logic of the product library that needs no GPU: what gamdp_align_batch settles before launching anything
(gamdp_task_preflight: the reference's pre-checks, banded_smith_waterman.cc:90-132, incl. windows that start past the
end of a) against the CPU oracle, and the product build's refusal to honour diagnostics switches."""
import os
import subprocess
import sys

import _cases
import _oracle as O
from gam_ngs_amd import api, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_preflight_agrees_with_the_oracle_on_every_case_it_settles():
    settled = {}
    n_launch = 0
    cases = _cases.beyond_cases(51, 2500, bands=(0, 1, 5, 20, 150, 512)) + _cases.cases(52, 2500, max_len=200)
    for c in cases:
        a, b = O.encode(c["a"]), O.encode(c["b"])
        st, cells = api.task_preflight(len(a), len(b), c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"], c["fs"], c["fe"])
        o, _ = O.oracle_align(a, b, c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"], c["fs"], c["fe"], want_ops=False)
        if st == lib.ST_OK:
            n_launch += 1
            assert cells == o.cells and o.status != O.INVALID, c   # it would launch: the reference sized a matrix
            # and the kernels are only ever handed windows that start inside the padded contig
            assert c["begin_a"] <= len(a) + c["band"]
        else:
            assert st == o.status, (c, st, o.status)
            assert cells == o.cells, (c, cells, o.cells)
            settled[st] = settled.get(st, 0) + 1
    assert n_launch > 1500
    assert settled.get(lib.ST_EMPTY, 0) > 100 and settled.get(lib.ST_OUT_OF_RANGE, 0) > 100, settled


def test_product_library_is_not_the_diagnostics_build():
    code = "import sys; sys.path.insert(0, %r); from gam_ngs_amd import lib; sys.exit(lib.load_library().gamdp_build_info())" % ROOT
    env = dict(os.environ)
    env.pop("GAMDP_LIB", None)
    assert subprocess.run([sys.executable, "-c", code], env=env).returncode == 0
    diag = os.path.join(ROOT, "gam_ngs_amd", "libgamdp_diag.so")
    assert os.path.exists(diag), "make -C gam_ngs_amd/csrc diag"
    assert subprocess.run([sys.executable, "-c", code], env=dict(env, GAMDP_LIB=diag)).returncode == 1
