"""CPU: the native text parsers of steps 2 and 3 (csrc/hostio/tsvscan.cpp, tsvstep3.cpp) under AddressSanitizer +
UndefinedBehaviorSanitizer against damaged tables (tests/native/fuzz_tsv.cpp): whatever is in the text, they return — a result, "not
mine" or an error — and never read outside the buffers they were given."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
H = os.path.join(ROOT, "longsom_amd", "csrc", "hostio")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_damaged_tables_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_tsv")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
                           os.path.join(ROOT, "tests", "native", "fuzz_tsv.cpp"), os.path.join(H, "tsvscan.cpp"), os.path.join(H, "tsvstep3.cpp"), os.path.join(H, "tsvwrite.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    for table, seed in (("sample.calling.step2.tsv", 1), ("sample.dist150.calling.step2.tsv", 2)):
        r = subprocess.run([exe, os.path.join(G, table), "400", str(seed)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        assert "fuzz_tsv:" in r.stdout and " 0 errors" in r.stdout, r.stdout
