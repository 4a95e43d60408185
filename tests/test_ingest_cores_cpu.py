"""CPU: the two cores of the device-side BAM ingest (csrc/inflate_core.h, csrc/bamrec_core.h) are host + device code; here they are
compiled for the host under AddressSanitizer + UndefinedBehaviorSanitizer and checked against zlib and against the host decoder
(tests/native/test_inflate.cpp, tests/native/test_bamrec.cpp).  The same code on the GPU: tests/test_ingest_gpu.py."""
import os
import shutil
import subprocess

import pytest

from longsom_amd import hostio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
SAN = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
       "-I" + os.path.join(ROOT, "longsom_amd", "csrc")]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


@pytest.mark.parametrize("wide_copy", [False, True])
def test_inflate_core_equals_zlib(tmp_path, wide_copy):
    """800 zlib streams (levels 0-9, every strategy, 0 - 65 536 bytes, the device's strided tables) inflate to zlib's bytes; 5 800
    truncated / bit-flipped ones are rejected or decoded without touching memory outside the buffers.  wide_copy: the match copy the
    DEVICE build uses (8 / 32 bytes a turn through unaligned words), compiled for the host."""
    exe = str(tmp_path / "test_inflate")
    subprocess.check_call(SAN + (["-DLSI_WIDE_COPY"] if wide_copy else []) + [os.path.join(ROOT, "tests", "native", "test_inflate.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe], env=ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "inflate ok" in r.stdout


@pytest.mark.parametrize("legacy", [0, 1])
def test_record_core_equals_the_host_decoder(tmp_path, legacy):
    """the reference-pinned multi-contig sample decoded with the device's record code (one lane and 64 emulated lanes) == the arrays,
    counters and per-barcode tallies of lsio_decode_bam, in both htslib modes of the 1D2D rule"""
    exe = str(tmp_path / "test_bamrec")
    subprocess.check_call(SAN + [os.path.join(ROOT, "tests", "native", "test_bamrec.cpp"), os.path.join(ROOT, "longsom_amd", "csrc", "hostio", "bamio.cpp"), "-o", exe, "-lz"])
    for stem in ("pileup.rand", "pileup.randsfx"):                   # (randsfx: the same reads with "-1"-suffixed CB tags)
        bc = hostio.read_barcodes(os.path.join(G, stem + ".barcodes.tsv"))
        names = tmp_path / (stem + ".barcodes.txt")
        names.write_text("".join(b + "\n" for b in bc.barcodes))
        path = os.path.join(G, stem + ".bam")
        for mq in (0, 30):
            r = subprocess.run([exe, path, str(names), str(mq), str(legacy)], env=ENV, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
