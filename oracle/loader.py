"""ctypes loader of oracle/liblongsom_oracle.so (built by oracle/Makefile).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblongsom_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.lso_count.restype = C.c_int64
    return _lib


def plp_set_legacy_del_merge(on):
    """htslib <= 1.10 behaviour of the BAM-level oracle's CIGAR step (see plp_oracle.c g_legacy_del_merge)"""
    lib().plp_set_legacy_del_merge(C.c_int(1 if on else 0))


def plp_count(bam_path, barcodes, celltype_of, ct, contig_len, refs, min_bq=20, min_mq=60, min_dp=5, min_cc=5, max_depth=200000, window=50000):
    """BAM-level column-major oracle (oracle/plp_oracle.c).  Returns keys, ref, counts[n,42].  window: the reference's pileup windows
    (--bin 50000): with a depth cap every window is a pileup of its own (BaseCellCounter.py:185-191)."""
    L = lib()
    L.plp_count.restype = C.c_int64
    contig_len = np.ascontiguousarray(contig_len, np.int64)
    celltype_of = np.ascontiguousarray(celltype_of, np.uint8)
    refs = [np.ascontiguousarray(r, np.uint8) for r in refs]
    ref_ptrs = (C.c_void_p * len(refs))(*[r.ctypes.data for r in refs])
    cap = int(sum(contig_len)) + 1
    keys = np.zeros(cap, np.int64); ref = np.zeros(cap, np.uint8); counts = np.zeros((cap, 42), np.uint32)
    n = L.plp_count(os.fsencode(bam_path), "\n".join(barcodes).encode(), C.c_int32(len(barcodes)), _p(celltype_of), C.c_int32(ct),
                    C.c_int32(len(contig_len)), _p(contig_len), ref_ptrs, C.c_int32(min_bq), C.c_int32(min_mq), C.c_int32(min_dp),
                    C.c_int32(min_cc), _p(keys), _p(ref), _p(counts), C.c_int64(cap), C.c_int32(max_depth), C.c_int32(window))
    if n < 0:
        raise RuntimeError("plp_count failed on %s" % bam_path)
    return keys[:n].copy(), ref[:n].copy(), counts[:n].copy()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def count(rec, contig_len, refs, celltype_of, ct, min_bq=20, min_mq=60, min_dp=5, min_cc=5,
          flag_exclude=0xF04, ignore_orphans=1, threads=1, region_w=512, cap=None, span=None):
    """Events-level oracle (oracle/count_oracle.c).  refs: list of uint8 arrays per contig.
    Returns keys, ref, counts[n,42], n_columns.  threads > 1: the region-parallel form (lso_count_mt, same per-column code).
    span = ((tid, pos), (tid, pos)): only the columns of that half-open region (lso_count_span_mt; one shard of a larger job)."""
    L = lib()
    L.lso_count_mt.restype = C.c_int64
    L.lso_count_span_mt.restype = C.c_int64
    contig_len = np.ascontiguousarray(contig_len, np.int64)
    celltype_of = np.ascontiguousarray(celltype_of, np.uint8)
    refs = [np.ascontiguousarray(r, np.uint8) for r in refs]
    ref_ptrs = (C.c_void_p * len(refs))(*[r.ctypes.data for r in refs])
    cap = int(cap) if cap else min(int(rec.n_events), int(sum(contig_len))) + 1
    while True:
        keys = np.empty(cap, np.int64); ref = np.empty(cap, np.uint8); counts = np.empty((cap, 42), np.uint32)
        ncols = C.c_int64(0)
        fn, extra = (L.lso_count, ()) if threads <= 1 else (L.lso_count_mt, (C.c_int32(int(threads)), C.c_int32(int(region_w))))
        if span is not None:
            (t0, p0), (t1, p1) = span
            fn, extra = L.lso_count_span_mt, (C.c_int32(max(1, int(threads))), C.c_int32(int(region_w)), C.c_int64((int(t0) << 32) | int(p0)), C.c_int64((int(t1) << 32) | int(p1)))
        n = fn(C.c_int64(rec.n_reads), C.c_int64(rec.n_segs), _p(rec.read_tid), _p(rec.read_flag), _p(rec.read_mapq),
                        _p(rec.read_cb), _p(rec.seg_read), _p(rec.seg_start), _p(rec.seg_len), _p(rec.seg_ev_off), _p(rec.events),
                        C.c_int32(len(contig_len)), _p(contig_len), ref_ptrs, _p(celltype_of), C.c_int32(len(celltype_of)), C.c_int32(ct),
                        C.c_int32(min_bq), C.c_int32(min_mq), C.c_int32(min_dp), C.c_int32(min_cc), C.c_uint32(flag_exclude),
                        C.c_int32(ignore_orphans), _p(keys), _p(ref), _p(counts), C.c_int64(cap), C.byref(ncols), *extra)
        if n < 0:
            raise MemoryError("oracle allocation failed")
        if n <= cap:
            return keys[:n].copy(), ref[:n].copy(), counts[:n].copy(), int(ncols.value)
        cap = int(n)
