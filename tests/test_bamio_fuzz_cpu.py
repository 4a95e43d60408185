"""CPU: the BAM reader under AddressSanitizer + UndefinedBehaviorSanitizer against damaged files (tests/native/fuzz_bamio.cpp):
truncated, bit-flipped and length-field-overwritten copies of a valid BAM must be decoded or rejected, never read out of bounds.
Also: the streaming form (lsio_stream_*) returns, batch by batch, exactly what one whole-file decode returns."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from longsom_amd import hostio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
SRC = os.path.join(ROOT, "longsom_amd", "csrc", "hostio", "bamio.cpp")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_damaged_bams_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_bamio")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-pthread",
                           os.path.join(ROOT, "tests", "native", "fuzz_bamio.cpp"), SRC, "-o", exe, "-lz"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    for seed in (1, 2):
        r = subprocess.run([exe, os.path.join(G, "pileup.rand.bam"), str(tmp_path), "150", str(seed)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-4000:]
        assert "rejected" in r.stdout


def test_truncated_and_corrupt_files_are_errors(tmp_path):
    raw = open(os.path.join(G, "pileup.rand.bam"), "rb").read()
    for name, data in (("cut.bam", raw[: len(raw) // 2]), ("cut2.bam", raw[:-40]), ("tiny.bam", raw[:10]), ("magic.bam", b"\x00" * 100)):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(RuntimeError):
            hostio.decode_bam(str(p), None)


@pytest.mark.parametrize("batch", [1, 3000, 70000, 1 << 20])
def test_stream_batches_equal_whole_decode(batch):
    bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
    bam = os.path.join(G, "pileup.rand.bam")
    whole = hostio.decode_bam(bam, bc.barcodes)
    parts = list(hostio.stream_bam(bam, bc.barcodes, batch_bytes=batch))
    assert len(parts) > (1 if batch < 1 << 20 else 0)
    rec = hostio.concat_records([p.records for p in parts])
    for name, _ in type(rec)._SPEC:
        assert np.array_equal(getattr(rec, name), getattr(whole.records, name)), name
    rep = {}
    for p in parts:
        for k, v in p.report.items():
            rep[k] = rep.get(k, 0) + v
    assert rep == whole.report
    assert parts[0].contig_names == whole.contig_names
    assert np.array_equal(sum(p.cb_pass for p in parts), whole.cb_pass) and np.array_equal(sum(p.cb_low for p in parts), whole.cb_low)
