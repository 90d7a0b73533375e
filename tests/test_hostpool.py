"""The host thread pool behind the parallel loops of a batch call (gam_ngs_amd/csrc/gamdp_hostpool.h) under ThreadSanitizer:
concurrent callers, back-to-back loops of every size, every element visited exactly once; and the launch planner's parallel radix sort
on top of it (gamdp_hostsort.h) against std::stable_sort.  CPU only (g++)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
@pytest.mark.parametrize("src", ["hostpool_test.cpp", "hostsort_test.cpp"])
def test_hostpool_under_sanitizers(tmp_path, san, src):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "hostpool_test")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-pthread", "-I", os.path.join(ROOT, "gam_ngs_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", src), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "bad 0" in r.stdout
