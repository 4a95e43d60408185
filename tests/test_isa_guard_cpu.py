"""CPU (cross-compile only): every 128-bit buffer store with an SGPR offset in the count kernels is followed by wait states before
anything else issues.  gfx950 reads such a store's data registers late and the compiler does not guard that form (DESIGN.md §3,
lessons); the guard is an inline s_nop kept in place by a scheduling barrier, and this test is what notices if a compiler or a
source change lets an instruction slip in between."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_wide_buffer_stores_are_followed_by_wait_states(tmp_path):
    out = tmp_path / "pileup.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                           os.path.join(ROOT, "longsom_amd", "csrc", "pileup.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", "//", "."))]
    wide = [i for i, l in enumerate(lines) if re.match(r"buffer_store_dwordx[34]\b", l)]
    assert len(wide) >= 9, "expected the row emission's quad stores"
    unguarded = []
    for i in wide:
        ops = lines[i].split(",")
        soffset = ops[-1].split()[0]                       # "... s[8:11], s67 offen" -> "s67";  "... s[8:11], 0 offen" -> "0"
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        if re.fullmatch(r"s\d+", soffset) and not nxt.startswith("s_nop"):
            unguarded.append((lines[i], nxt))
    assert not unguarded, "128-bit buffer store with an SGPR offset not followed by s_nop: %s" % (unguarded[:3],)
