"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (longsom_hip.h: the boundary; longsom_synth.h: measurement and
test support), and the ctypes binding declares the same set (no compute call: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("longsom_hip.h", "longsom_synth.h")]
LIB = os.path.join(ROOT, "longsom_amd", "lib", "liblongsom_hip.so")
IO_LIB = os.path.join(ROOT, "longsom_amd", "lib", "liblongsom_io.so")


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", "\n".join(open(h).read() for h in HEADERS), flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|void|int64_t)\s+(lsg_[a-z0-9_]+)\s*\(", text, flags=re.M)
    assert len(names) >= 25
    return sorted(set(names))


def test_the_boundary_header_declares_nothing_synthetic():
    """what a rule's script binds (include/longsom_hip.h) has no generator and no read-back of resident arrays in it"""
    text = re.sub(r"/\*.*?\*/", "", open(HEADERS[0]).read(), flags=re.S)
    assert "lsg_synth" not in text and "lsg_copy_reads_to_host" not in text


def test_header_symbols_are_exported():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in the header but not exported: %s" % missing


def test_binding_covers_the_header():
    from longsom_amd import _lib
    declared = set(declared_functions())
    bound = set(_lib.SIGNATURES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_no_device_is_an_error_not_a_fallback():
    """without a GPU the product path fails loudly (lsg_create) instead of computing on the host"""
    import torch
    if torch.cuda.is_available() or torch.cuda.device_count() > 0 or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    from longsom_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(0)


def test_host_io_library_exports():
    lib = ctypes.CDLL(IO_LIB)
    for n in ("lsio_decode_bam", "lsio_free_decoded", "lsio_split_bam", "lsio_synth_bam", "lsio_synth_records", "lsio_barcode", "lsio_ref_bases",
              "lsio_write_count_rows", "lsio_write_merged_rows", "lsio_write_step1_rows", "lsio_free_text", "lsio_last_error", "lsio_tsv_last_error"):
        assert hasattr(lib, n), n
