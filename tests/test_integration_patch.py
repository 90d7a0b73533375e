"""integration/gam-merge-gamdp.patch: the adapter for the real gam-merge.

* The bridge translation unit -- the only file of the patch that calls the C ABI -- compiles against include/gamdp.h
  in isolation (it is free of GAM-NGS / Boost types on purpose).
* Where the reference tree is available (the build container, not the GPU box): the patch applies cleanly to a copy
  of it, the new files it adds are the master copies under integration/, and the patched tree's bridge compiles.
The patched gam-merge cannot be linked here (Boost, sparsehash absent): applying + type-checking is the bar."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "gam-merge-gamdp.patch")


def test_bridge_compiles_against_the_c_abi_alone(tmp_path):
    inc = tmp_path / "inc" / "pctg"
    inc.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "integration", "GamdpBridge.hpp"), inc)
    for std in ("gnu++98", "gnu++17"):   # the reference is C++98-style code that also builds as C++17
        r = subprocess.run(["g++", "-std=" + std, "-Wall", "-Wextra", "-Werror", "-c", "-I" + os.path.join(ROOT, "include"),
                            "-I" + str(tmp_path / "inc"), os.path.join(ROOT, "integration", "GamdpBridge.cc"), "-o",
                            str(tmp_path / "bridge.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    # every gamdp_* symbol the bridge uses is exported by the library
    nm = subprocess.run(["nm", "-u", str(tmp_path / "bridge.o")], capture_output=True, text=True).stdout
    used = sorted(set(l.split()[-1] for l in nm.splitlines() if " gamdp_" in l))
    assert "gamdp_multi_align_merge_blocks" in used and "gamdp_multi_seqset_create" in used
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "gam_ngs_amd", "libgamdp.so"))
    for name in used:
        assert hasattr(lib, name), name


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "src", "pctg")), reason="reference tree not present")
def test_patch_applies_cleanly_to_the_reference(tmp_path):
    tree = tmp_path / "gam-ngs"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns(".git", "bamtools-2.3.0"))
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tree, capture_output=True, text=True)
    assert dry.returncode == 0 and "FAILED" not in dry.stdout and "fuzz" not in dry.stdout, dry.stdout + dry.stderr
    real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tree, capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    for rel, master in (("lib/include/pctg/GamdpBridge.hpp", "GamdpBridge.hpp"), ("lib/src/pctg/GamdpBridge.cc", "GamdpBridge.cc")):
        assert open(tree / rel).read() == open(os.path.join(ROOT, "integration", master)).read(), rel
    r = subprocess.run(["g++", "-std=gnu++11", "-Wall", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-Ilib/include",
                        "lib/src/pctg/GamdpBridge.cc"], cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the seam and the three-phase run are where the patch says they are
    bf = open(tree / "lib/src/pctg/BuildPctgFunctions.cc").read()
    assert "builder.alignMergeBlock(graph,*mb);" in bf and "prepareMergeLists" in bf and "finishPctg" in bf
    tb = open(tree / "lib/src/pctg/ThreadedBuildPctg.cc").read()
    assert "ThreadedBuildPctg::runOnGpu()" in tb and "if( gamdp_bridge::ready() ) return this->runOnGpu();" in tb
    assert "gamdp_bridge::init( masterCodes, slaveCodes );" in open(tree / "src/Merge.cc").read()
    assert "GamdpBridge.cc" in open(tree / "CMakeLists.txt").read()


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "src", "pctg")), reason="reference tree not present")
def test_make_l1_dump_script_first_half(tmp_path):
    """integration/make_l1_dump.sh up to where a Boost host is needed: a patched copy of the reference, the stand-in
    library for hosts without hipcc (exports what the bridge binds, refuses every call), the bridge compiled and linked
    against it with no undefined symbol.  The rest (cmake, gam-merge with GAMDP_DUMP_PREFIX, the copy into
    tests/golden/l1_reference_dump/) cannot run here."""
    env = dict(os.environ, TMPDIR=str(tmp_path), GAMDP_FORCE_STUB="1")
    r = subprocess.run(["bash", os.path.join(ROOT, "integration", "make_l1_dump.sh"), REF, "selftest", str(tmp_path / "no_example"), "--check-only"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "bridge compiles and links" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert not os.path.exists(os.path.join(ROOT, "tests", "golden", "l1_reference_dump", "selftest"))
    # the stand-in refuses: the bridge then leaves gam-merge on its CPU path
    so = tmp_path / "stub.so"
    assert subprocess.run(["gcc", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-o", str(so),
                           os.path.join(ROOT, "integration", "gamdp_stub.c")]).returncode == 0
    import ctypes
    stub = ctypes.CDLL(str(so))
    h = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert stub.gamdp_multi_create(dev, 1, ctypes.byref(h)) != 0 and not h.value
