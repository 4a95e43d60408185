"""Host-side handle on the HIP hot path (one Engine per GPU).

Mirrors the stages of LongSom's SNV chain as methods:
  set_barcodes   <- meta_to_dict / split_bam routing   (scripts/PreProcessing/SplitBamCellTypes.py:16-36,83-90)
  load_reads     <- reading the BAM                     (scripts/SNVCalling/BaseCellCounter.py:190-191)
  pileup_count   <- run_interval over all windows       (BaseCellCounter.py:182-320)
  call_step1     <- merge + variant_calling_step1       (MergeBaseCellCounts.py:116-204, BaseCellCalling.step1.py:19-476)
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import CallParams, CountParams, CountStats, Reads, ROW_WORDS


@dataclass
class ReadRecords:
    """Pre-decoded read-record arrays (SoA form of a coordinate-sorted BAM; include/longsom_hip.h)."""
    read_tid: np.ndarray      # int32 [R]
    read_pos: np.ndarray      # int32 [R]
    read_flag: np.ndarray     # uint16 [R]
    read_mapq: np.ndarray     # uint8 [R]
    read_cb: np.ndarray       # int32 [R]  dense barcode id, -1 = none
    seg_read: np.ndarray      # uint32 [S]
    seg_start: np.ndarray     # int32 [S]
    seg_len: np.ndarray       # int32 [S]
    seg_ev_off: np.ndarray    # int64 [S]
    events: np.ndarray        # uint16 [E]  LSG_EVENT: 0x0800 | sym << 8 | qual, 0 for 'NA'
    read_names: Optional[List[str]] = field(default=None, repr=False)

    _SPEC = (("read_tid", np.int32), ("read_pos", np.int32), ("read_flag", np.uint16), ("read_mapq", np.uint8),
             ("read_cb", np.int32), ("seg_read", np.uint32), ("seg_start", np.int32), ("seg_len", np.int32),
             ("seg_ev_off", np.int64), ("events", np.uint16))

    def __post_init__(self):
        for name, dt in self._SPEC:
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        assert len(self.read_pos) == len(self.read_tid) == len(self.read_flag) == len(self.read_mapq) == len(self.read_cb)
        assert len(self.seg_start) == len(self.seg_read) == len(self.seg_len) == len(self.seg_ev_off)

    @classmethod
    def empty(cls) -> "ReadRecords":
        return cls(*[np.zeros(0, dt) for _, dt in cls._SPEC])

    @property
    def n_reads(self): return len(self.read_tid)
    @property
    def n_segs(self): return len(self.seg_read)
    @property
    def n_events(self): return len(self.events)

    def subset(self, read_mask: np.ndarray) -> "ReadRecords":
        """Records of the reads selected by a boolean mask (events re-packed)."""
        read_mask = np.asarray(read_mask, dtype=bool)
        new_idx = np.cumsum(read_mask) - 1
        seg_keep = read_mask[self.seg_read]
        seg_len = self.seg_len[seg_keep]
        seg_off_old = self.seg_ev_off[seg_keep]
        seg_off_new = np.concatenate([[0], np.cumsum(seg_len)[:-1]]).astype(np.int64) if len(seg_len) else np.zeros(0, np.int64)
        if len(seg_len):
            idx = np.repeat(seg_off_old - seg_off_new, seg_len) + np.arange(int(seg_len.sum()), dtype=np.int64)
            events = self.events[idx]
        else:
            events = np.zeros(0, np.uint16)
        names = [n for n, m in zip(self.read_names, read_mask) if m] if self.read_names is not None else None
        return ReadRecords(self.read_tid[read_mask], self.read_pos[read_mask], self.read_flag[read_mask],
                           self.read_mapq[read_mask], self.read_cb[read_mask],
                           new_idx[self.seg_read[seg_keep]].astype(np.uint32), self.seg_start[seg_keep], seg_len,
                           seg_off_new, events, names)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One liblongsom_hip handle.  Raises if the HIP library or a GPU is missing (no CPU fallback)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, keep_reads: bool = False):
        """keep_reads: loads also keep the compact events beside the tile store, so that reads_to_host() can return them
        (tests, sampling for the CPU baseline); the product paths leave it off — the store is the only resident copy."""
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.lsg_create(int(device), C.byref(h)), "lsg_create")
        self._h = h
        self.device = device
        self.n_ct = 0
        self.n_contigs = 0
        self.contig_len = None
        self.pileup_window = 50000
        self._load_settings = {"load_filter": (0, 0, 0), "count_at_load": None, "store_policy": 0, "keep_unlisted": False}      # the library's defaults
        if stream is not None:
            self.set_stream(stream)
        if keep_reads:
            self.set_keep_reads(True)

    def set_load_filter(self, min_mq: int = 0, flag_exclude: int = 0, ignore_orphans: int = 0):
        """Reads failing these are not stored by the next loads (what SplitBamCellTypes.py:110-113 does before BaseCellCounter sees the
        BAM); later counts must be at least as strict.  Default: no filter (lsg_set_load_filter)."""
        _lib.check(self._lib.lsg_set_load_filter(self._h, int(min_mq), int(flag_exclude), int(ignore_orphans)), "lsg_set_load_filter")
        self._load_settings["load_filter"] = (int(min_mq), int(flag_exclude), int(ignore_orphans))

    LAYOUT_COMPACT, LAYOUT_PHASED = 0, 1

    def set_events_layout(self, layout: int = 0):
        """what the caller promises about the events of the next load_reads calls: LAYOUT_PHASED = every segment at an offset congruent to its
        reference start modulo 128 (lsg_set_events_layout; checked by the load, which falls back by itself)"""
        _lib.check(self._lib.lsg_set_events_layout(self._h, int(layout)), "lsg_set_events_layout")

    def set_pileup_window(self, window: int = 50000):
        """the reference's pileup windows (BaseCellCounter.py --bin): the loads that follow cut their entries at the window edges, the depth
        cap of a count is replayed per window (lsg_set_pileup_window)"""
        _lib.check(self._lib.lsg_set_pileup_window(self._h, int(window)), "lsg_set_pileup_window")
        self.pileup_window = int(window)

    def set_count_at_load(self, params=None):
        """The loads that follow also make the first count under `params` (a CountParams), in the pass that builds the store; the
        first pileup_count(params) after such a load returns that count without another pass.  None switches it off
        (lsg_set_count_at_load).  Barcodes, references and region must be set before the load."""
        import ctypes as C
        _lib.check(self._lib.lsg_set_count_at_load(self._h, C.byref(params) if params is not None else None), "lsg_set_count_at_load")
        self._load_settings["count_at_load"] = params

    def set_keep_unlisted(self, on: bool):
        """The BAM loads that follow keep reads without a listed barcode (cb = -1): never counted, but part of the buffer the per-cell
        genotyping's pileup of the unsplit BAM fills (HCCVSingleCellGenotype.py:121-122; lsg_set_keep_unlisted)."""
        _lib.check(self._lib.lsg_set_keep_unlisted(self._h, 1 if on else 0), "lsg_set_keep_unlisted")
        self._load_settings["keep_unlisted"] = bool(on)

    STORE_KEEP, STORE_SKIP_WHEN_COUNTED = 0, 1

    def set_store_policy(self, policy: int):
        """STORE_SKIP_WHEN_COUNTED: a load that makes its count in its own pass (set_count_at_load) counts straight from the caller's
        events and keeps no tile store - for a BAM that is counted once (the reference's BaseCellCounter rule); anything that needs
        the store afterwards (another count, genotype_cells) raises until reads are loaded again.  STORE_KEEP is the default
        (lsg_set_store_policy)."""
        _lib.check(self._lib.lsg_set_store_policy(self._h, int(policy)), "lsg_set_store_policy")
        self._load_settings["store_policy"] = int(policy)

    def load_settings(self) -> dict:
        """what the next loads will do, as this wrapper last set it (load filter, count at load, store policy, unlisted reads): a caller that
        changes them for one load puts them back with restore_load_settings"""
        return dict(self._load_settings)

    def restore_load_settings(self, saved: dict):
        self.set_load_filter(*saved["load_filter"])
        self.set_count_at_load(saved["count_at_load"])
        self.set_store_policy(saved["store_policy"])
        self.set_keep_unlisted(saved["keep_unlisted"])

    def unload_reads(self):
        """give the device memory of the resident load (reads, store, rows, call records, cached temporaries) back: lsg_unload_reads"""
        _lib.check(self._lib.lsg_unload_reads(self._h), "lsg_unload_reads")
        self._n_rows = [0] * max(1, self.n_ct)
        self._n_sites = 0

    def set_keep_reads(self, keep: bool):
        _lib.check(self._lib.lsg_set_keep_reads(self._h, 1 if keep else 0), "lsg_set_keep_reads")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self): return self
    def __exit__(self, *exc): self.close()

    # ---- inputs ------------------------------------------------------------------------------
    def set_stream(self, hip_stream: int):
        _lib.check(self._lib.lsg_set_stream(self._h, C.c_void_p(hip_stream)), "lsg_set_stream")

    def synchronize(self):
        _lib.check(self._lib.lsg_synchronize(self._h), "lsg_synchronize")

    def set_contigs(self, lengths):
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        _lib.check(self._lib.lsg_set_contigs(self._h, len(lengths), _ptr(lengths)), "lsg_set_contigs")
        self.n_contigs = len(lengths)
        self.contig_len = lengths.copy()

    def load_reference(self, tid: int, bases, on_device: bool = False, length: Optional[int] = None):
        """bases: uint8 array of upper-cased ASCII bases (host), or a device pointer + length."""
        if on_device:
            _lib.check(self._lib.lsg_load_reference(self._h, tid, C.c_void_p(int(bases)), int(length), 1), "lsg_load_reference")
            return
        if isinstance(bases, (bytes, bytearray, str)):
            bases = np.frombuffer(bases.encode() if isinstance(bases, str) else bytes(bases), dtype=np.uint8)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        _lib.check(self._lib.lsg_load_reference(self._h, tid, _ptr(bases) if bases.size else C.c_void_p(bases.ctypes.data), len(bases), 0),
                   "lsg_load_reference")

    def set_barcodes(self, celltype_of, n_celltypes: int):
        ct = np.ascontiguousarray(celltype_of, dtype=np.uint8)
        # (a table identical to the one the handle holds is left alone by the library itself: setting one drops the resident count, and the one
        # a load made under set_count_at_load would be counted again)
        _lib.check(self._lib.lsg_set_barcodes(self._h, _ptr(ct), len(ct), int(n_celltypes)), "lsg_set_barcodes")
        self.n_ct = int(n_celltypes)
        self.n_cb = len(ct)

    def load_reads(self, rec: ReadRecords):
        """Builds the tile store of these reads on the device (lsg_load_reads); the arrays are free again on return."""
        r = Reads(rec.n_reads, rec.n_segs, rec.n_events, *[_ptr(getattr(rec, n)) for n, _ in ReadRecords._SPEC], 0)
        _lib.check(self._lib.lsg_load_reads(self._h, C.byref(r)), "lsg_load_reads")

    def load_reads_device(self, n_reads, n_segs, n_events, ptrs: dict):
        """ptrs: name -> device pointer (int) for every array of ReadRecords._SPEC; arrays stay owned by the caller."""
        r = Reads(n_reads, n_segs, n_events, *[C.c_void_p(int(ptrs[n])) for n, _ in ReadRecords._SPEC], 1)
        _lib.check(self._lib.lsg_load_reads(self._h, C.byref(r)), "lsg_load_reads")

    def load_reads_struct(self, reads: Reads):
        """lsg_load_reads of a ready lsg_reads (e.g. what synth_generate returned)"""
        _lib.check(self._lib.lsg_load_reads(self._h, C.byref(reads)), "lsg_load_reads")

    def load_bam(self, path: str, barcodes, min_mapq: int = 60, first_record_offset: Optional[int] = None, legacy_del_merge: Optional[bool] = None):
        """Device-side ingest (lsg_load_bam): the BAM's bytes go to the GPU as they are; BGZF inflate, record decode, CB lookup, SplitBam's
        counters and the CIGAR walk run there and the tile store is built from the device arrays.  Set the contigs (hostio.bam_header)
        first.  Returns (info dict with the report counters and the phases' times, cb_pass, cb_low per dense barcode id).  Raises
        _lib.LsgError with rc -4 in its text ("straddle") for a BAM whose records are not aligned to its BGZF blocks: decode that one on
        the host (hostio.decode_bam)."""
        import mmap
        from . import hostio
        from ._lib import BamInfo
        if first_record_offset is None:
            first_record_offset = hostio.bam_header(path)[2]
        if legacy_del_merge is None:
            legacy_del_merge = bool(hostio.load().lsio_get_legacy_del_merge())
        joined = "\n".join(barcodes).encode()
        n_cb = len(barcodes)
        info = BamInfo()
        cb_pass = np.zeros(n_cb, np.int64); cb_low = np.zeros(n_cb, np.int64)
        with open(path, "rb") as f:
            size = os.fstat(f.fileno()).st_size
            with mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
                buf = np.frombuffer(mm, dtype=np.uint8)
                try:
                    _lib.check(self._lib.lsg_load_bam(self._h, C.c_void_p(buf.ctypes.data), size, int(first_record_offset), joined, n_cb, None, int(min_mapq),
                                                      1 if legacy_del_merge else 0, C.byref(info), _ptr(cb_pass), _ptr(cb_low), n_cb), "lsg_load_bam")
                finally:
                    del buf
        d = {k: getattr(info, k) for k, _ in BamInfo._fields_ if k != "pad_"}
        return d, cb_pass, cb_low

    def load_bam_range(self, slice_bytes: np.ndarray, first_record_offset: int, barcodes, min_mapq: int, count_lo_key: int, count_hi_key: int,
                       legacy_del_merge: Optional[bool] = None):
        """lsg_load_bam_range: a slice of whole BGZF blocks of a coordinate-sorted BAM (uint8 array, e.g. a view of a memory map), a record
        starting at first_record_offset of its first block; the counters and tallies take the records with (tid << 32 | pos) in
        [count_lo_key, count_hi_key).  Returns what load_bam returns; info["last_key"] = key of the slice's last complete record."""
        from . import hostio
        from ._lib import BamInfo
        if legacy_del_merge is None:
            legacy_del_merge = bool(hostio.load().lsio_get_legacy_del_merge())
        joined = "\n".join(barcodes).encode()
        n_cb = len(barcodes)
        info = BamInfo()
        cb_pass = np.zeros(n_cb, np.int64); cb_low = np.zeros(n_cb, np.int64)
        buf = np.ascontiguousarray(slice_bytes, dtype=np.uint8)
        _lib.check(self._lib.lsg_load_bam_range(self._h, C.c_void_p(buf.ctypes.data), len(buf), int(first_record_offset), joined, n_cb, None, int(min_mapq),
                                                1 if legacy_del_merge else 0, int(count_lo_key), int(count_hi_key), C.byref(info), _ptr(cb_pass), _ptr(cb_low), n_cb),
                   "lsg_load_bam_range")
        return {k: getattr(info, k) for k, _ in BamInfo._fields_ if k != "pad_"}, cb_pass, cb_low

    def set_region(self, tid_lo=0, pos_lo=0, tid_hi=None, pos_hi=0):
        """Count only columns in [(tid_lo,pos_lo), (tid_hi,pos_hi)) — window sharding across GPUs."""
        tid_hi = self.n_contigs if tid_hi is None else tid_hi
        _lib.check(self._lib.lsg_set_region(self._h, int(tid_lo), int(pos_lo), int(tid_hi), int(pos_hi)), "lsg_set_region")

    # ---- synthetic workload (bench / tests) -------------------------------------------------------
    def synth_reference(self, seed: int):
        _lib.check(self._lib.lsg_synth_reference(self._h, C.c_uint64(seed)), "lsg_synth_reference")

    def synth_reads(self, model):
        """model: longsom_amd.synth.SynthModel; generates its read-record arrays in HBM and loads them."""
        mc = model.as_c()
        _lib.check(self._lib.lsg_synth_reads(self._h, C.byref(mc)), "lsg_synth_reads")

    def synth_generate(self, model) -> Reads:
        """Only generates the model's compact read-record arrays in HBM (owned by the handle until the next generate / synth_reads)
        and returns their lsg_reads: a stand-in for a caller with a device-resident decoded BAM (bench.py times load_reads on it)."""
        mc = model.as_c()
        out = Reads()
        _lib.check(self._lib.lsg_synth_generate(self._h, C.byref(mc), C.byref(out)), "lsg_synth_generate")
        return out

    def reads_shape(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.lsg_get_reads_shape(self._h, C.byref(a), C.byref(b), C.byref(c)), "lsg_get_reads_shape")
        return int(a.value), int(b.value), int(c.value)

    def reads_to_host(self, events: bool = True) -> ReadRecords:
        """Copy the resident read-record arrays back to the host (tests, CPU-baseline sampling); the events need keep_reads.
        events=False: the per-read and per-segment arrays only (always resident; events and seg_ev_off come back empty / zero)."""
        R, S, E = self.reads_shape()
        arrs = {n: np.zeros({"read": R, "seg_": S, "even": E if events else 0}[n[:4]], dt) for n, dt in ReadRecords._SPEC}
        ptrs = [(_ptr(arrs[n]) if events or n not in ("events", "seg_ev_off") else None) for n, _ in ReadRecords._SPEC]
        r = Reads(R, S, E, *ptrs, 0)
        _lib.check(self._lib.lsg_copy_reads_to_host(self._h, C.byref(r)), "lsg_copy_reads_to_host")
        return ReadRecords(**arrs)

    def reference_to_host(self, tid: int) -> np.ndarray:
        out = np.zeros(int(self.contig_len[tid]), np.uint8)
        if out.size:
            _lib.check(self._lib.lsg_copy_reference_to_host(self._h, int(tid), _ptr(out)), "lsg_copy_reference_to_host")
        return out

    # ---- hot path ----------------------------------------------------------------------------
    def pileup_count(self, params: Optional[CountParams] = None):
        """Returns (rows per cell type, pileup columns with >= 1 counted entry summed over cell types)."""
        params = params or CountParams.longsom_defaults()
        n_rows = (C.c_int64 * _lib.MAX_CELLTYPES)()
        n_cols = C.c_int64(0)
        _lib.check(self._lib.lsg_pileup_count(self._h, C.byref(params), n_rows, C.byref(n_cols)), "lsg_pileup_count")
        self._n_rows = [int(n_rows[i]) for i in range(self.n_ct)]
        return list(self._n_rows), int(n_cols.value)

    def fetch_counts(self, ct: int):
        """Rows of one cell type in genomic order: keys int64 (tid<<32|pos0), ref uint8, counts uint32 [n,42]."""
        n = self._n_rows[ct]
        keys = np.empty(n, np.int64); ref = np.empty(n, np.uint8); counts = np.empty((n, ROW_WORDS), np.uint32)
        if n:
            _lib.check(self._lib.lsg_fetch_counts(self._h, ct, _ptr(keys), _ptr(ref), _ptr(counts), n), "lsg_fetch_counts")
        return keys, ref, counts

    def load_counts(self, keys_per_ct, counts_per_ct):
        """Install per-cell-type count rows (parsed from BaseCellCounter TSVs) instead of running the pileup."""
        n_ct = len(keys_per_ct)
        ks = [np.ascontiguousarray(k, dtype=np.int64) for k in keys_per_ct]
        cs = [np.ascontiguousarray(c, dtype=np.uint32).reshape(-1, ROW_WORDS) for c in counts_per_ct]
        kp = (C.c_void_p * n_ct)(*[k.ctypes.data for k in ks])
        cp = (C.c_void_p * n_ct)(*[c.ctypes.data for c in cs])
        nr = (C.c_int64 * n_ct)(*[len(k) for k in ks])
        _lib.check(self._lib.lsg_load_counts(self._h, n_ct, kp, cp, nr), "lsg_load_counts")
        self.n_ct = n_ct
        self._n_rows = [len(k) for k in ks]

    def max_live_reads(self) -> int:
        """Upper bound on the reads simultaneously live in the reference's pileup buffer (see lsg_max_live_reads)."""
        v = int(self._lib.lsg_max_live_reads(self._h))
        _lib.check(-1 if v < 0 else 0, "lsg_max_live_reads")
        return v

    def count_stats(self) -> CountStats:
        s = CountStats()
        _lib.check(self._lib.lsg_get_count_stats(self._h, C.byref(s)), "lsg_get_count_stats")
        return s

    def layout_info(self):
        """(path, build_ms, store_bytes): path 2 = the load built the store alone, 3 = in the pass that also made the first count (set_count_at_load), 4 = the load counted and kept no store (set_store_policy); wall time the last load spent building the
        store; device bytes the store and what belongs to it hold (lsg_get_layout_info)"""
        path = C.c_int32(0); ms = C.c_double(0.0); nbytes = C.c_int64(0)
        _lib.check(self._lib.lsg_get_layout_info(self._h, C.byref(path), C.byref(ms), C.byref(nbytes)), "lsg_get_layout_info")
        return int(path.value), float(ms.value), int(nbytes.value)

    def build_times(self):
        """HIP-event ms of the last load's build: capacities + scatter, sort, per-entry words, event gather (lsg_get_build_times)"""
        ms = (C.c_float * 4)()
        _lib.check(self._lib.lsg_get_build_times(self._h, ms), "lsg_get_build_times")
        return [float(x) for x in ms]

    def store_shape(self):
        """(entries, 1 KB blocks, events) of the resident tile store (lsg_get_store_shape)"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.lsg_get_store_shape(self._h, C.byref(a), C.byref(b), C.byref(c)), "lsg_get_store_shape")
        return int(a.value), int(b.value), int(c.value)

    def max_live_reads_all(self) -> int:
        """the bound of max_live_reads over every resident read with a barcode, whatever its cell type (lsg_max_live_reads_all)"""
        v = int(self._lib.lsg_max_live_reads_all(self._h))
        _lib.check(-1 if v < 0 else 0, "lsg_max_live_reads_all")
        return v

    def max_live_reads_exact(self) -> int:
        """per cell type, per POSITION: the largest buffer a pushed read can meet, itself included - a count with max_depth >= this
        drops nothing (lsg_max_live_reads_exact)"""
        v = int(self._lib.lsg_max_live_reads_exact(self._h))
        _lib.check(-1 if v < 0 else 0, "lsg_max_live_reads_exact")
        return v

    def call_step1(self, params: Optional[CallParams] = None):
        params = params or CallParams.longsom_defaults()
        n_sites = C.c_int64(0); n_cand = C.c_int64(0)
        _lib.check(self._lib.lsg_call_step1(self._h, C.byref(params), C.byref(n_sites), C.byref(n_cand)), "lsg_call_step1")
        self._n_sites = int(n_sites.value); self._n_cand = int(n_cand.value)
        return self._n_sites, self._n_cand

    def fetch_calls(self, candidates_only: bool = False):
        cap = self._n_sites
        arr = np.empty(max(cap, 1), np.dtype(_lib.Call))
        n_out = C.c_int64(0)
        _lib.check(self._lib.lsg_fetch_calls(self._h, C.c_void_p(arr.ctypes.data), cap, 1 if candidates_only else 0, C.byref(n_out)), "lsg_fetch_calls")
        return arr[:int(n_out.value)]

    def export_calls(self, kind: int, dst_device_ptr: int = 0, capacity: int = 0) -> int:
        """Compact call records into a caller-owned device buffer (kind: 0 all, 1 step-2 rows, 2 PASS
        candidates); dst 0 = count only.  Returns the number of selected rows."""
        n_out = C.c_int64(0)
        _lib.check(self._lib.lsg_export_calls(self._h, int(kind), C.c_void_p(int(dst_device_ptr)) if dst_device_ptr else None,
                                              int(capacity), C.byref(n_out)), "lsg_export_calls")
        return int(n_out.value)

    # ---- the tables' text, printed on the device (include/longsom_hip.h: enum lsg_table) ----
    TABLE_COUNTS, TABLE_MERGED, TABLE_STEP1, TABLE_STEP1_KEPT, TABLE_STEP2, TABLE_STEP3_ROWS = 0, 4, 5, 6, 7, 8

    def set_table_names(self, contig_names, celltype_names) -> None:
        """Names the rows print (contigs in set_contigs order, cell types in index order)."""
        _lib.check(self._lib.lsg_set_table_names(self._h, len(contig_names), "\n".join(contig_names).encode(), len(celltype_names),
                                                 "\n".join(celltype_names).encode()), "lsg_set_table_names")

    def format_table(self, table: int) -> int:
        """Prints the rows of one table (TABLE_COUNTS + cell type, TABLE_MERGED, TABLE_STEP1, TABLE_STEP1_KEPT) into a device buffer the
        engine keeps; returns the size of the text."""
        n = C.c_int64(0)
        _lib.check(self._lib.lsg_format_table(self._h, int(table), C.byref(n)), "lsg_format_table")
        return int(n.value)

    def table_bytes(self, table: int, n_bytes: int, prefix: bytes = b"") -> bytes:
        """prefix + the formatted text as one bytes object."""
        api = C.pythonapi
        api.PyBytes_FromStringAndSize.restype = C.py_object
        api.PyBytes_FromStringAndSize.argtypes = [C.c_void_p, C.c_ssize_t]
        api.PyBytes_AsString.restype = C.c_void_p
        api.PyBytes_AsString.argtypes = [C.py_object]
        out = api.PyBytes_FromStringAndSize(None, len(prefix) + n_bytes)      # (uninitialised: filled below before anybody else sees it)
        at = api.PyBytes_AsString(out)
        if prefix:
            C.memmove(at, prefix, len(prefix))
        _lib.check(self._lib.lsg_copy_table(self._h, int(table), C.c_void_p(at + len(prefix)), n_bytes), "lsg_copy_table")
        return out

    def append_table(self, table: int, path: str) -> None:
        """Appends the formatted text to `path`; may run on a thread of its own beside other calls (not beside format_table / free_table
        of the same table)."""
        import os
        _lib.check(self._lib.lsg_append_table(self._h, int(table), os.fsencode(path)), "lsg_append_table")

    def step2_summary(self, n_cols: int):
        """After format_table(TABLE_STEP2): (kinds of cell per column over all its rows - the bits of tsvio.column_kinds -, size of the rows
        step 3 can keep, which are table TABLE_STEP3_ROWS now)."""
        kinds = np.zeros(int(n_cols), np.uint8)
        n = C.c_int64(0)
        _lib.check(self._lib.lsg_step2_summary(self._h, int(n_cols), _ptr(kinds), C.byref(n)), "lsg_step2_summary")
        return kinds, int(n.value)

    def free_table(self, table: int = -1) -> None:
        _lib.check(self._lib.lsg_free_table(self._h, int(table)), "lsg_free_table")

    def load_posset(self, kind: int, keys, on_device: bool = False, n: Optional[int] = None):
        if on_device:
            _lib.check(self._lib.lsg_load_posset(self._h, kind, C.c_void_p(int(keys)), int(n), 1), "lsg_load_posset")
            return
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        _lib.check(self._lib.lsg_load_posset(self._h, kind, _ptr(keys), len(keys), 0), "lsg_load_posset")

    def genotype_cells(self, site_keys, alt_sym, params=None):
        """Per-cell (Dp, Alt) at target sites (HCCVSingleCellGenotype.py:82-220).  site_keys = (tid << 32) | pos0, strictly
        ascending; alt_sym = expected alt symbol class per site.  Returns two uint32 arrays [n_sites, n_cb]."""
        from ._lib import GenotypeParams
        params = params or GenotypeParams.longsom_defaults()
        site_keys = np.ascontiguousarray(site_keys, dtype=np.int64)
        alt_sym = np.ascontiguousarray(alt_sym, dtype=np.uint8)
        assert len(site_keys) == len(alt_sym)
        dp = np.zeros((len(site_keys), self.n_cb), np.uint32)
        alt = np.zeros((len(site_keys), self.n_cb), np.uint32)
        if len(site_keys):
            _lib.check(self._lib.lsg_genotype_cells(self._h, C.byref(params), len(site_keys), _ptr(site_keys), _ptr(alt_sym), _ptr(dp), _ptr(alt), 0),
                       "lsg_genotype_cells")
        return dp, alt

    def genotype_cells_grouped(self, site_keys, alt_sym, group_off, params=None, max_depth: int = 200000):
        """genotype_cells with the pileup's depth cap replayed per window of target sites (lsg_genotype_cells_grouped): the sites
        [group_off[g], group_off[g + 1]) are one pileup call of the reference (HCCVSingleCellGenotype.py:109-122)."""
        from ._lib import GenotypeParams
        params = params or GenotypeParams.longsom_defaults()
        site_keys = np.ascontiguousarray(site_keys, dtype=np.int64)
        alt_sym = np.ascontiguousarray(alt_sym, dtype=np.uint8)
        group_off = np.ascontiguousarray(group_off, dtype=np.int64)
        assert len(site_keys) == len(alt_sym)
        dp = np.zeros((len(site_keys), self.n_cb), np.uint32)
        alt = np.zeros((len(site_keys), self.n_cb), np.uint32)
        if len(site_keys):
            _lib.check(self._lib.lsg_genotype_cells_grouped(self._h, C.byref(params), int(max_depth), len(site_keys), _ptr(site_keys), _ptr(alt_sym),
                                                            len(group_off) - 1, _ptr(group_off), _ptr(dp), _ptr(alt), 0), "lsg_genotype_cells_grouped")
        return dp, alt

    def betabinom_sf4(self, k, n, alpha: float, beta: float) -> np.ndarray:
        """round(betabinom.sf(k - 0.001, n, alpha, beta), 4) * 1e4 as int32, evaluated on the device."""
        k = np.ascontiguousarray(k, dtype=np.uint32); n = np.ascontiguousarray(n, dtype=np.uint32)
        out = np.zeros(len(k), np.int32)
        if len(k):
            _lib.check(self._lib.lsg_betabinom_sf4(self._h, len(k), _ptr(k), _ptr(n), float(alpha), float(beta), _ptr(out)), "lsg_betabinom_sf4")
        return out

    def betabinom_sf(self, k, n, alpha: float, beta: float):
        """(round(p, 4) * 1e4 as int32, p as float64) with p = betabinom.sf(k - 0.001, n, alpha, beta) evaluated on the device"""
        k = np.ascontiguousarray(k, dtype=np.uint32); n = np.ascontiguousarray(n, dtype=np.uint32)
        out = np.zeros(len(k), np.int32); raw = np.zeros(len(k), np.float64)
        if len(k):
            _lib.check(self._lib.lsg_betabinom_sf(self._h, len(k), _ptr(k), _ptr(n), float(alpha), float(beta), _ptr(out), _ptr(raw)), "lsg_betabinom_sf")
        return out, raw

    def probe_posset_device(self, kind: int, keys_ptr: int, n: int, hits_ptr: int) -> None:
        """membership of n device-resident int64 keys in set `kind`, hits (uint8) written to device memory (both caller-owned)"""
        _lib.check(self._lib.lsg_probe_posset(self._h, kind, C.c_void_p(int(keys_ptr)), int(n), C.c_void_p(int(hits_ptr)), 1), "lsg_probe_posset")

    def probe_posset(self, kind: int, keys) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        hits = np.zeros(len(keys), np.uint8)
        if len(keys):
            _lib.check(self._lib.lsg_probe_posset(self._h, kind, _ptr(keys), len(keys), _ptr(hits), 0), "lsg_probe_posset")
        return hits
