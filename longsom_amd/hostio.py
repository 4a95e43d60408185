"""ctypes binding of liblongsom_io.so (longsom_amd/csrc/hostio/bamio.cpp): BAM -> read-record arrays,
barcodes.tsv -> barcode table, synthetic BAM / FASTA writer.

  read_barcodes   <- meta_to_dict   workflow/scripts/PreProcessing/SplitBamCellTypes.py:16-36
  decode_bam      <- split_bam's record loop (:65-124) + the CIGAR walk of pysam's pileup
"""
import ctypes as C
import os
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from .engine import ReadRecords

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblongsom_io.so")
_lib = None


class Decoded(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("n_segs", C.c_int64), ("n_events", C.c_int64),
        ("read_tid", C.c_void_p), ("read_pos", C.c_void_p), ("read_flag", C.c_void_p), ("read_mapq", C.c_void_p), ("read_cb", C.c_void_p),
        ("seg_read", C.c_void_p), ("seg_start", C.c_void_p), ("seg_len", C.c_void_p), ("seg_ev_off", C.c_void_p), ("events", C.c_void_p),
        ("n_contigs", C.c_int32), ("contig_names", C.c_char_p), ("contig_len", C.c_void_p),
        ("total_reads", C.c_int64), ("pass_reads", C.c_int64), ("cb_not_found", C.c_int64), ("cb_not_matched", C.c_int64),
        ("mapq_filtered", C.c_int64),
        ("n_barcodes", C.c_int32), ("barcodes", C.c_char_p),
        ("n_tally", C.c_int64), ("cb_pass", C.c_void_p), ("cb_low", C.c_void_p),
    ]


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C longsom_amd/csrc/hostio`")
        lib = C.CDLL(LIB_PATH)
        lib.lsio_last_error.restype = C.c_char_p
        lib.lsio_decode_bam.restype = C.c_int
        lib.lsio_decode_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.POINTER(Decoded))]
        lib.lsio_free_decoded.argtypes = [C.POINTER(Decoded)]
        lib.lsio_stream_open.restype = C.c_int
        lib.lsio_stream_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.lsio_stream_next.restype = C.c_int
        lib.lsio_stream_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(Decoded))]
        lib.lsio_stream_close.argtypes = [C.c_void_p]
        lib.lsio_synth_bam.restype = C.c_int
        lib.lsio_synth_bam.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.lsio_synth_records.restype = C.c_int
        lib.lsio_synth_records.argtypes = [C.c_void_p, C.POINTER(C.POINTER(Decoded))]
        lib.lsio_barcode.argtypes = [C.c_uint64, C.c_int64, C.c_char_p]
        lib.lsio_split_bam.restype = C.c_int
        lib.lsio_split_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p]
        lib.lsio_build_bai.restype = C.c_int
        lib.lsio_build_bai.argtypes = [C.c_char_p, C.c_char_p]
        lib.lsio_set_legacy_del_merge.argtypes = [C.c_int]
        lib.lsio_get_legacy_del_merge.restype = C.c_int
        lib.lsio_set_keep_unlisted.argtypes = [C.c_int]
        lib.lsio_get_keep_unlisted.restype = C.c_int
        _lib = lib
        if os.environ.get("LONGSOM_HTSLIB_LEGACY_DEL_MERGE", "0") == "1":
            lib.lsio_set_legacy_del_merge(1)
    return _lib


def set_legacy_del_merge(on: bool) -> bool:
    """htslib <= 1.10 compatibility of the CIGAR -> column step (process-wide; returns the previous setting).  Before htslib 1.11,
    bam_plp's resolve_cigar2 flagged the last column of ANY operation followed by a deletion — also of a D operation itself — so inside
    "1D2D" the first deletion's column is printed "*-2NN" by pysam and counted as 'D' by EasyReadPileup (BaseCellCounter.py:167-170),
    where htslib >= 1.11 (the default here) merges consecutive D's and leaves that column 'O'.  The reference's conda environment pins
    neither pysam nor htslib (workflow/envs/SComatic.yaml:9,23).  Also settable with LONGSOM_HTSLIB_LEGACY_DEL_MERGE=1."""
    lib = load()
    old = bool(lib.lsio_get_legacy_del_merge())
    lib.lsio_set_legacy_del_merge(1 if on else 0)
    return old


def set_keep_unlisted(on: bool) -> bool:
    """The host decoder keeps reads without a CB tag or with an unlisted barcode (cb = -1) instead of dropping them (process-wide; returns
    the previous setting).  They are never counted, but the per-cell genotyping piles up the UNSPLIT BAM (HCCVSingleCellGenotype.py:121-122)
    and its max_depth buffer holds every read that passes the pileup's own filters.  Engine.set_keep_unlisted is the device decoder's twin."""
    lib = load()
    old = bool(lib.lsio_get_keep_unlisted())
    lib.lsio_set_keep_unlisted(1 if on else 0)
    return old


def _err(what):
    raise RuntimeError("%s: %s" % (what, load().lsio_last_error().decode("utf-8", "replace")))


@dataclass
class BarcodeTable:
    """barcodes.tsv as the device sees it: cleaned barcode -> dense id -> cell-type index."""
    barcodes: List[str]            # cleaned, unique, dense id = position
    celltype_of: np.ndarray        # uint8 [n_cb]
    celltype_names: List[str]      # index -> name, python-sorted (fixed column order)


def read_barcodes(path: str) -> BarcodeTable:
    """meta_to_dict: tab-separated with header, columns Index and Cell_type; Index loses everything from the
    first '-' (regex '-.*$', :20), spaces in Cell_type become '_' (:24); duplicated cleaned barcodes: the last
    row wins (dict semantics of to_dict, :31)."""
    import pandas as pd
    meta = pd.read_csv(path, delimiter="\t")
    clean = [re.sub("-.*$", "", str(x)) for x in meta["Index"]]
    ctype = [str(x).replace(" ", "_") for x in meta["Cell_type"]]
    mapping: Dict[str, str] = {}
    for b, c in zip(clean, ctype):
        mapping[b] = c
    names = sorted(set(mapping.values()))
    idx = {n: i for i, n in enumerate(names)}
    barcodes = list(mapping)
    return BarcodeTable(barcodes, np.asarray([idx[mapping[b]] for b in barcodes], np.uint8), names)


class _DecodedOwner:
    """Keeps a lsio_decoded block alive for as long as any array that views its memory exists."""
    def __init__(self, lib, handle):
        self._lib, self._handle = lib, handle

    def __del__(self):
        try:
            self._lib.lsio_free_decoded(self._handle)
        except Exception:
            pass


def _take(d: Decoded, owner: "_DecodedOwner") -> ReadRecords:
    """The decoder's arrays as numpy views (no copy: at BAM scale the copies cost as much as the decode); every view holds the
    owner through its ctypes buffer, the C memory is freed when the last of them goes away."""
    def arr(ptr, n, dt):
        if n == 0 or not ptr:
            return np.zeros(0, dt)
        buf = (C.c_char * (int(n) * np.dtype(dt).itemsize)).from_address(ptr)
        buf._owner = owner
        return np.frombuffer(buf, dtype=dt)
    R, S, E = d.n_reads, d.n_segs, d.n_events
    return ReadRecords(arr(d.read_tid, R, np.int32), arr(d.read_pos, R, np.int32), arr(d.read_flag, R, np.uint16), arr(d.read_mapq, R, np.uint8),
                       arr(d.read_cb, R, np.int32), arr(d.seg_read, S, np.uint32), arr(d.seg_start, S, np.int32), arr(d.seg_len, S, np.int32),
                       arr(d.seg_ev_off, S, np.int64), arr(d.events, E, np.uint16))


@dataclass
class DecodedBam:
    records: ReadRecords
    contig_names: List[str]
    contig_len: np.ndarray
    report: Dict[str, int]         # the counters of SplitBamCellTypes' report.txt (:62,117-124)
    barcodes: Optional[List[str]] = None   # auto-barcode mode: the distinct cleaned CBs found (dense id = index)
    cb_pass: Optional[np.ndarray] = None   # listed-barcode mode: per barcode id, matched reads with MAPQ >= min_mapq ...
    cb_low: Optional[np.ndarray] = None    # ... and below it (for the report of a re-annotated barcode table)

    def report_for(self, keep: np.ndarray) -> Dict[str, int]:
        """SplitBamCellTypes' report had the barcode table listed only the barcodes where `keep` is set (re-annotation pass 2)."""
        keep = np.asarray(keep, bool)
        dropped = int(self.cb_pass[~keep].sum() + self.cb_low[~keep].sum())
        rep = {"Total_reads": self.report["Total_reads"], "Pass_reads": int(self.cb_pass[keep].sum()), "CB_not_found": self.report["CB_not_found"],
               "CB_not_matched": self.report["CB_not_matched"] + dropped}
        low = int(self.cb_low[keep].sum())
        if low:
            rep["MAPQ"] = low
        return rep


def _wrap(lib, out, barcodes) -> DecodedBam:
    owner = _DecodedOwner(lib, out)
    d = out.contents
    rec = _take(d, owner)
    names = d.contig_names.decode().split("\n")[: d.n_contigs] if d.n_contigs else []
    lens = np.ctypeslib.as_array(C.cast(d.contig_len, C.POINTER(C.c_int64)), shape=(d.n_contigs,)).copy() if d.n_contigs else np.zeros(0, np.int64)
    rep = {"Total_reads": d.total_reads, "Pass_reads": d.pass_reads, "CB_not_found": d.cb_not_found, "CB_not_matched": d.cb_not_matched}
    if d.mapq_filtered:
        rep["MAPQ"] = d.mapq_filtered
    found = d.barcodes.decode().split("\n")[: d.n_barcodes] if barcodes is None else None
    tally = lambda ptr: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(len(barcodes),)).copy() if barcodes is not None and d.n_tally >= len(barcodes) and len(barcodes) else None
    return DecodedBam(rec, names, lens, rep, found, tally(d.cb_pass), tally(d.cb_low))


def bam_header(path: str):
    """(contig names, contig lengths int64, length of the header in the uncompressed stream = offset of the first record): the BAM
    header parsed on the host — the first BGZF block(s) only — for the device-side ingest (lsg_load_bam), which needs the reference
    table before it starts (pysam.AlignmentFile's header, SplitBamCellTypes.py:51)."""
    import struct
    import zlib
    data = b""
    with open(path, "rb") as f:
        def more():
            nonlocal data
            h = f.read(18)
            if len(h) < 18 or h[:3] != b"\x1f\x8b\x08" or not (h[3] & 4):
                raise RuntimeError("%s is not BGZF" % path)
            xlen = struct.unpack_from("<H", h, 10)[0]
            extra = h[12:18] + f.read(xlen - 6)
            bsize, q = None, 0
            while q + 4 <= xlen:
                slen = struct.unpack_from("<H", extra, q + 2)[0]
                if extra[q:q + 2] == b"BC" and slen == 2:
                    bsize = struct.unpack_from("<H", extra, q + 4)[0] + 1
                q += 4 + slen
            if bsize is None or bsize < xlen + 20:
                raise RuntimeError("%s: corrupt BGZF block" % path)
            body = f.read(bsize - xlen - 12)
            if len(body) != bsize - xlen - 12:
                raise RuntimeError("%s: truncated BGZF block" % path)
            data += zlib.decompress(body[:-8], -15)

        def need(n):
            while len(data) < n:
                more()
        need(12)
        if data[:4] != b"BAM\x01":
            raise RuntimeError("%s has no BAM magic" % path)
        p = 8 + struct.unpack_from("<I", data, 4)[0]
        need(p + 4)
        n_ref = struct.unpack_from("<I", data, p)[0]; p += 4
        names, lens = [], []
        for _ in range(n_ref):
            need(p + 4)
            l_name = struct.unpack_from("<I", data, p)[0]; p += 4
            need(p + l_name + 4)
            names.append(data[p:p + l_name].split(b"\0")[0].decode()); p += l_name
            lens.append(struct.unpack_from("<I", data, p)[0]); p += 4
    return names, np.asarray(lens, np.int64), p


def stream_bam(path: str, barcodes: Optional[Sequence[str]], min_mapq: int = 60, threads: int = 0, batch_bytes: int = 1 << 30):
    """The BAM in file order, a batch at a time (about batch_bytes of uncompressed BAM each): yields DecodedBam objects whose
    records, counters and per-barcode tallies cover one batch.  The compressed file is mapped, a batch is inflated and decoded by
    `threads` host threads while the caller works on the previous one (the ctypes call releases the GIL).  This is the reference's
    window-by-window reading (BaseCellCounter.py:81-113,190-191) without an index: file order is coordinate order."""
    lib = load()
    h = C.c_void_p()
    joined = None if barcodes is None else "\n".join(barcodes).encode()
    if lib.lsio_stream_open(os.fsencode(path), joined, -1 if barcodes is None else len(barcodes), None, int(min_mapq), int(threads), C.byref(h)) != 0:
        _err("lsio_stream_open")
    try:
        while True:
            out = C.POINTER(Decoded)()
            rc = lib.lsio_stream_next(h, int(batch_bytes), C.byref(out))
            if rc < 0:
                _err("lsio_stream_next")
            if rc == 0:
                return
            yield _wrap(lib, out, barcodes)
    finally:
        lib.lsio_stream_close(h)


def concat_records(parts: Sequence[ReadRecords]) -> ReadRecords:
    """Read-record arrays of several batches as one (read and event indices re-based)."""
    parts = [p for p in parts if p.n_reads or p.n_segs]
    if not parts:
        z = lambda dt: np.zeros(0, dt)
        return ReadRecords(*[z(dt) for _, dt in ReadRecords._SPEC])
    if len(parts) == 1:
        return parts[0]
    r0 = np.cumsum([0] + [p.n_reads for p in parts]); e0 = np.cumsum([0] + [p.n_events for p in parts])
    cat = lambda n: np.concatenate([getattr(p, n) for p in parts])
    return ReadRecords(cat("read_tid"), cat("read_pos"), cat("read_flag"), cat("read_mapq"), cat("read_cb"),
                       np.concatenate([p.seg_read.astype(np.int64) + r0[i] for i, p in enumerate(parts)]).astype(np.uint32), cat("seg_start"), cat("seg_len"),
                       np.concatenate([p.seg_ev_off + e0[i] for i, p in enumerate(parts)]), cat("events"))


def decode_bam(path: str, barcodes: Optional[Sequence[str]], min_mapq: int = 60, threads: int = 0) -> DecodedBam:
    """BAM -> read-record arrays.  Reads without a CB tag or whose cleaned CB is not in `barcodes` are dropped
    (they can never be counted); flags and MAPQ are kept for the device-side admission.  barcodes=None: every
    distinct cleaned CB of the file is a cell (how BaseCellCounter sees a per-cell-type BAM)."""
    lib = load()
    out = C.POINTER(Decoded)()
    if barcodes is None:
        rc = lib.lsio_decode_bam(os.fsencode(path), None, -1, None, int(min_mapq), int(threads), C.byref(out))
    else:
        joined = "\n".join(barcodes).encode()
        rc = lib.lsio_decode_bam(os.fsencode(path), joined, len(barcodes), None, int(min_mapq), int(threads), C.byref(out))
    if rc != 0:
        _err("lsio_decode_bam")
    return _wrap(lib, out, barcodes)


def split_bam(path: str, table: "BarcodeTable", out_paths: Sequence[str], min_mapq: int = 60) -> Dict[str, int]:
    """SplitBamCellTypes' BAM outputs: one BAM per cell type (out_paths in table.celltype_names order); returns the
    report counters."""
    lib = load()
    ct = np.ascontiguousarray(table.celltype_of, np.uint8)
    cnt = (C.c_int64 * 5)()
    if lib.lsio_split_bam(os.fsencode(path), "\n".join(table.barcodes).encode(), len(table.barcodes), ct.ctypes.data_as(C.c_void_p),
                          len(table.celltype_names), "\n".join(out_paths).encode(), int(min_mapq), cnt) != 0:
        _err("lsio_split_bam")
    rep = {"Total_reads": cnt[0], "Pass_reads": cnt[1], "CB_not_found": cnt[2], "CB_not_matched": cnt[3]}
    if cnt[4]:
        rep["MAPQ"] = cnt[4]
    return rep


def synth_barcodes(model) -> List[str]:
    lib = load()
    buf = C.create_string_buffer(17)
    res = []
    for cb in range(model.n_cb):
        lib.lsio_barcode(C.c_uint64(model.seed), cb, buf)
        res.append(buf.value.decode())
    return res


def ref_bases(seed: int, tid: int, length: int) -> np.ndarray:
    """Reference bases of one contig of the synthetic genome (host twin of Engine.synth_reference)."""
    lib = load()
    out = np.empty(int(length), np.uint8)
    lib.lsio_ref_bases.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_void_p]
    if length:
        lib.lsio_ref_bases(C.c_uint64(seed), int(tid), int(length), out.ctypes.data_as(C.c_void_p))
    return out


def synth_bam(model, bam_path: str, fasta_path: Optional[str] = None, barcode_suffix: str = "") -> None:
    """Write the model's reads as a coordinate-sorted BAM (+ the synthetic reference as FASTA)."""
    lib = load()
    mc = model.as_c()
    names = "\n".join(model.contig_names).encode()
    lens = np.ascontiguousarray(model.contig_len, np.int64)
    if lib.lsio_synth_bam(C.byref(mc), names, lens.ctypes.data_as(C.c_void_p), os.fsencode(bam_path),
                          os.fsencode(fasta_path) if fasta_path else None, barcode_suffix.encode() if barcode_suffix else None) != 0:
        _err("lsio_synth_bam")


def synth_records(model) -> ReadRecords:
    """Host evaluation of the model (same arrays the GPU generator produces)."""
    lib = load()
    out = C.POINTER(Decoded)()
    mc = model.as_c()
    if lib.lsio_synth_records(C.byref(mc), C.byref(out)) != 0:
        _err("lsio_synth_records")
    return _take(out.contents, _DecodedOwner(lib, out))


def write_barcodes_tsv(path: str, barcodes: Sequence[str], celltype_of, celltype_names: Sequence[str], suffix: str = "") -> None:
    with open(path, "w") as f:
        f.write("Index\tCell_type\n")
        for b, c in zip(barcodes, celltype_of):
            f.write("%s%s\t%s\n" % (b, suffix, celltype_names[int(c)]))


# ---- the BAM index (.bai): linear index only ---------------------------------------------------------------------------------------
def build_bai(bam: str, bai: Optional[str] = None) -> str:
    """writes <bam>.bai with the linear index of a coordinate-sorted BAM (lsio_build_bai; no binning index — what samtools index
    writes has both, read_bai takes either).  For the synthetic and fixture BAMs: real data comes with its .bai (rules/SNVCalling.smk:6-7)."""
    lib = load()
    bai = bai or bam + ".bai"
    if lib.lsio_build_bai(os.fsencode(bam), os.fsencode(bai)) != 0:
        raise RuntimeError(lib.lsio_last_error().decode("utf-8", "replace"))
    return bai


def find_bai(bam: str) -> Optional[str]:
    """the BAM's index if there is one that is not older than the BAM (samtools warns about such a pair; here it is not used: a slice
    cut by a stale index would be counted without any sign of trouble)"""
    import sys
    for p in (bam + ".bai", os.path.splitext(bam)[0] + ".bai"):
        if os.path.exists(p):
            if os.path.getmtime(p) + 1.0 < os.path.getmtime(bam):
                sys.stderr.write("warning: %s is older than %s: the index is not used (the whole file is ingested)\n" % (p, bam))
                return None
            return p
    return None


def read_bai(path: str) -> List[np.ndarray]:
    """per reference, the linear index of a .bai (SAM specification 5.2): ioffset[w] = smallest virtual offset (block start << 16 |
    offset inside the block) of the alignments overlapping the 16 kb window w, 0 = none recorded.  The binning index is skipped."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"BAI\x01":
        raise ValueError("%s is not a BAI file" % path)
    (n_ref,) = struct.unpack_from("<i", data, 4)
    at = 8
    out = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", data, at); at += 4
        for _ in range(n_bin):
            _, n_chunk = struct.unpack_from("<Ii", data, at); at += 8 + 16 * n_chunk
        (n_intv,) = struct.unpack_from("<i", data, at); at += 4
        out.append(np.frombuffer(data, dtype="<u8", count=n_intv, offset=at).copy()); at += 8 * n_intv
        if at > len(data):
            raise ValueError("%s is truncated" % path)
    return out
