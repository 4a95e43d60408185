"""GPU: lsg_export_calls (compaction of call records into a caller-owned device buffer) against lsg_fetch_calls filtered on the
host, for the three kinds; kind 2 (PASS candidates: what travels between GPUs) both through the list the call stage leaves behind
and through the generic scan."""
import ctypes as C

import numpy as np
import pytest

from longsom_amd import _lib, synth
from longsom_amd._lib import CallParams, CountParams

pytestmark = pytest.mark.gpu
SF_CANDIDATE, CF_PASS = 1 << 31, 6


class DeviceBuffer:
    """hipMalloc through the HIP runtime the library itself is linked to (torch brings a runtime of its own: initialising it in a
    process where the library's is already up finds no device)."""
    def __init__(self, nbytes):
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert self.hip.hipMemset(self.ptr, 0, C.c_size_t(nbytes)) == 0
        self.nbytes = nbytes

    def to_host(self):
        out = (C.c_uint8 * self.nbytes)()
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out, self.ptr, C.c_size_t(self.nbytes), 2) == 0      # hipMemcpyDeviceToHost
        return bytes(out)

    def free(self):
        self.hip.hipFree(self.ptr)


def export(engine, kind):
    n = engine.export_calls(kind)
    buf = DeviceBuffer(max(n, 1) * C.sizeof(_lib.Call))
    try:
        assert engine.export_calls(kind, buf.ptr.value, max(n, 1)) == n
        return np.frombuffer(buf.to_host(), dtype=np.dtype(_lib.Call), count=n)
    finally:
        buf.free()


def test_export_kinds_equal_filtered_fetch(engine, monkeypatch):
    m = synth.named("C1", n_reads=60000, n_genes=20, n_cb=120, snp_mod=97)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    engine.pileup_count(CountParams.longsom_defaults())
    # lenient thresholds so that the synthetic SNPs (shared by both cell types) come out as PASS candidates
    engine.call_step1(CallParams.longsom_defaults(min_ac_cells=1, min_ac_reads=2, min_cell_types=1, max_cell_types=2))
    calls = engine.fetch_calls()
    assert len(calls) > 1000
    sf = calls["site_filter"]
    is_pass = (sf == SF_CANDIDATE) & (calls["ct_filter"] == CF_PASS).any(axis=1)
    want = {0: calls, 1: calls[sf != 0], 2: calls[is_pass]}
    assert 3 <= len(want[2]) < len(want[1]) < len(want[0])
    for kind in (0, 1, 2):
        got = export(engine, kind)
        assert got.tobytes() == want[kind].tobytes(), kind
    monkeypatch.setenv("LSG_NO_PASS_LIST", "1")                   # the scan path of kind 2 (taken when the list overflows)
    assert export(engine, 2).tobytes() == want[2].tobytes()
    # a tighter call makes the list shorter, possibly empty: the count must follow
    monkeypatch.delenv("LSG_NO_PASS_LIST")
    engine.call_step1(CallParams.longsom_defaults(min_ac_cells=50, min_ac_reads=500, min_cell_types=1, max_cell_types=2))
    calls = engine.fetch_calls()
    is_pass = (calls["site_filter"] == SF_CANDIDATE) & (calls["ct_filter"] == CF_PASS).any(axis=1)
    assert export(engine, 2).tobytes() == calls[is_pass].tobytes()


def test_capacity_error(engine):
    n = engine.export_calls(1)
    if n < 2:
        pytest.skip("needs at least two rows")
    buf = DeviceBuffer(C.sizeof(_lib.Call))
    try:
        with pytest.raises(RuntimeError, match="capacity"):
            engine.export_calls(1, buf.ptr.value, 1)
    finally:
        buf.free()
