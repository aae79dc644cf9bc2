"""The container / header parsers of the product under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not
available on the pool): mutated zkey and wtns files -- bit flips, truncations, section sizes near 2^32, 2^63 and 2^64 --
must end in a normal return or a C++ exception, never in an out-of-bounds access (tests/native/fuzz_parsers.cpp). The
.wtns buffer of a proving service is untrusted input; the reference's reader accepts a section size that wraps its cursor
(src/binfile_utils.cpp:60-66)."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "ultragroth_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("fuzz") / "fuzz_parsers")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC,
           os.path.join(ROOT, "tests", "native", "fuzz_parsers.cpp"), os.path.join(CSRC, "host_util.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this g++ has no sanitizer runtime: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("name,kind", [("circuit_final.zkey", "zkey"), ("witness.wtns", "wtns"),
                                       (os.path.join("trapdoor", "ultra.uwtns"), "wtns")])
def test_mutated_files_never_crash_the_parsers(harness, name, kind):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2, 3):
        r = subprocess.run([harness, os.path.join(GOLDEN, name), kind, "4000", str(seed)], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        assert r.stdout.strip().endswith("0 crashed")
        parsed, rejected = int(r.stdout.split()[0]), int(r.stdout.split()[2])
        assert rejected > 1000 and parsed + rejected == 4000          # the mutations do bite, and every run is accounted for
