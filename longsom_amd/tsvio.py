"""Text side of the rule contract: the TSVs LongSom's SNV rules exchange.

Writers reproduce, byte for byte (except the wall-clock ##fileDate line):
  BaseCellCounter output      workflow/scripts/SNVCalling/BaseCellCounter.py:54-61,297-312 (rows in
                              python-string chromosome order then position, :64-70)
  MergeBaseCellCounts output  workflow/scripts/SNVCalling/MergeBaseCellCounts.py:59-86,134-137,170-172
  BaseCellCalling.step1 out   workflow/scripts/SNVCalling/BaseCellCalling.step1.py:48-76,398-401,464-467
Parsers read the same files back into the arrays the C-ABI takes (lsg_load_counts).
"""
import os
import time
from typing import List, Sequence

import numpy as np

from ._lib import ROW_WORDS

ALLELES = ["A", "C", "T", "G", "I", "D", "N", "O"]
INFO_FIELD = "DP|NC|CC|BC|BQ|BCf|BCr"
_CONCEPTS = (
    '##INFO=DP,Description="Depth of coverage">\n'
    '##INFO=NC,Description="Number of different cells">\n'
    '##INFO=CC,Description="Cell counts [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
    '##INFO=BC,Description="Base counts [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
    '##INFO=BQ,Description="Base quality sums [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
    '##INFO=BCf,Description="Base counts in forward reads [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
    '##INFO=BCr,Description="Base counts in reverse reads [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n')

STEP1_INFO_LINES = [
    "##INFO=ALT,Description=Alternative alleles found",
    "##INFO=FILTER,Description=Filter status of the variant site",
    "##INFO=Cell_types,Description=Cell type/s with the variant",
    "##INFO=Up_context,Description=Up-stream bases in reference (4 bases)",
    "##INFO=Down_context,Description=Down-stream bases in reference (4 bases)",
    "##INFO=N_ALT,Description=Cell type/s with the variant",
    "##INFO=Dp,Description=Depth of coverage (reads) in the cell type supporting the variant",
    "##INFO=Nc,Description=Number of distinct cells found in the cell type with the mutation",
    "##INFO=Bc,Description=Number of reads (base count) supporting the variants in the cell type with the mutation",
    "##INFO=Cc,Description=Number of distinct cells supporting the variant in the cell type with the mutation",
    "##INFO=VAF,Description=Variant allele frequency of variant in the cell type with the mutation",
    "##INFO=MCF,Description=Cancer cell fraction (fraction of ditinct cells) supporting the alternative allele in the cell type with the mutation",
    "##INFO=BCp,Description=Beta-binomial p-value for the variant allele (considering read counts)",
    "##INFO=CCp,Description=Beta-binomial p-value for the variant allele (considering cell counts)",
    "##INFO=Cell_types_min_BC,Description=Number of cell types with a minimum number of reads covering a site",
    "##INFO=Cell_types_min_CC,Description=Number of cell types with a minimum number of distinct cells found in a specific site",
    "##INFO=Rest_BC,Description=Base counts (reads) supporting other alternative alleles in this site. BC;DP;P-value (betabin)",
    "##INFO=Rest_CC,Description=Cell counts supporting other alternative alleles in this site. CC;NC;P-value (betabin)",
    "##INFO=Fisher_p,Description=Strand bias test. Fisher exact test p-value between forward and reverse reads in variant and reference allele",
    "##INFO=Cell_type_Filter,Description=Filter status of the variant site in each cell type",
]
STEP1_COLUMNS = ["ALT", "FILTER", "Cell_types", "Up_context", "Down_context", "N_ALT", "Dp", "Nc", "Bc", "Cc", "VAF", "MCF", "BCp",
                 "CCp", "Cell_types_min_BC", "Cell_types_min_CC", "Rest_BC", "Rest_CC", "Fisher_p", "Cell_type_Filter"]
CT_FILTER_NAMES = ["", "Non-Significant", "Low-Significance", "Multi-allelic", "Low_cells", "Low_reads", "PASS"]
SITE_FILTER_NAMES = [(1, "Multiple_cell_types"), (2, "Multi-allelic"), (4, "Min_cell_types"), (8, "Cell_type_noise"),
                     (16, "Noisy_site"), (32, "LC_Upstream"), (64, "LC_Downstream")]
SF_CANDIDATE = 1 << 31


def file_date() -> str:
    return "##fileDate=%s\n" % time.strftime("%d/%m/%Y")


def chrom_order(contig_names: Sequence[str]) -> np.ndarray:
    """rank[tid] of each contig in python string order (BaseCellCounter.py:64, MergeBaseCellCounts.py:111)."""
    order = sorted(range(len(contig_names)), key=lambda i: contig_names[i])
    rank = np.empty(len(contig_names), np.int64)
    rank[order] = np.arange(len(contig_names))
    return rank


def sort_like_reference(keys: np.ndarray, contig_names: Sequence[str]) -> np.ndarray:
    """permutation putting (tid,pos) keys into the reference's (chrom string, position) order"""
    rank = chrom_order(contig_names)
    tid = (keys >> 32).astype(np.int64)
    pos = keys & 0xFFFFFFFF
    return np.lexsort((pos, rank[tid]))


def row_text(c: np.ndarray) -> str:
    """'DP|NC|CC|BC|BQ|BCf|BCr' value string of one 42-word row (6 printed classes, BaseCellCounter.py:300-308)."""
    j = lambda o: ":".join(map(str, c[o:o + 6].tolist()))
    return "%d|%d|%s|%s|%s|%s|%s" % (c[0], c[1], j(2), j(10), j(18), j(26), j(34))


def format_counts_tsv(keys, refs, counts, contig_names, sample_id, date_line=None) -> str:
    out = [date_line or file_date(), _CONCEPTS, "\t".join(["#CHROM", "POS", "REF", "INFO", str(sample_id)]) + "\n"]
    perm = sort_like_reference(keys, contig_names)
    for i in perm.tolist():
        k = int(keys[i])
        out.append("%s\t%d\t%s\t%s\t%s\n" % (contig_names[k >> 32], (k & 0xFFFFFFFF) + 1, chr(int(refs[i])), INFO_FIELD, row_text(counts[i])))
    return "".join(out)


def parse_counts_tsv(path, contig_names):
    """BaseCellCounter TSV -> keys int64 (tid<<32|pos0) ascending in (tid,pos), refs uint8, counts uint32 [n,42], sample id."""
    tid_of = {n: i for i, n in enumerate(contig_names)}
    keys, refs, rows = [], [], []
    sample_id = None
    with open(path) as f:
        for line in f:
            if line.startswith("##"):
                continue
            if line.startswith("#CHROM"):
                sample_id = line.rstrip("\n").split("\t")[-1]
                continue
            line = line.rstrip("\n")
            if not line:
                continue
            chrom, pos, ref, _info, data = line.split("\t")
            dp, nc, cc, bc, bq, bcf, bcr = data.split("|")
            r = np.zeros(ROW_WORDS, np.uint32)
            r[0] = int(dp); r[1] = int(nc)
            for off, vec in ((2, cc), (10, bc), (18, bq), (26, bcf), (34, bcr)):
                v = [int(x) for x in vec.split(":")]
                r[off:off + len(v)] = v
            keys.append((tid_of[chrom] << 32) | (int(pos) - 1)); refs.append(ord(ref[0])); rows.append(r)
    keys = np.asarray(keys, np.int64); refs = np.asarray(refs, np.uint8)
    counts = np.stack(rows) if rows else np.zeros((0, ROW_WORDS), np.uint32)
    perm = np.argsort(keys, kind="stable")
    return keys[perm], refs[perm], counts[perm], sample_id


def format_merged_tsv(per_ct, contig_names, celltype_names, date_line=None) -> str:
    """per_ct: list of (keys, refs, counts) in (tid,pos) order.  Outer join on the site, 'NA' where a cell
    type has no row (MergeBaseCellCounts.py:71-84)."""
    out = [date_line or file_date(), _CONCEPTS, "\t".join(["#CHROM", "Start", "End", "REF", "INFO"] + list(celltype_names)) + "\n"]
    all_keys = np.unique(np.concatenate([k for k, _, _ in per_ct])) if per_ct else np.zeros(0, np.int64)
    idx = [dict(zip(k.tolist(), range(len(k)))) for k, _, _ in per_ct]
    perm = sort_like_reference(all_keys, contig_names)
    for i in perm.tolist():
        k = int(all_keys[i])
        cols, ref = [], None
        refs_seen = []
        for ct, (keys, refs, counts) in enumerate(per_ct):
            j = idx[ct].get(k)
            if j is None:
                cols.append("NA")
            else:
                cols.append(row_text(counts[j])); refs_seen.append(chr(int(refs[j])))
        # sort_set (MergeBaseCellCounts.py:48-57): distinct REFs by decreasing count, first-seen order on ties
        ref = "|".join(sorted(dict.fromkeys(refs_seen), key=lambda r: -refs_seen.count(r)))
        p1 = (k & 0xFFFFFFFF) + 1
        out.append("%s\t%d\t%d\t%s\t%s\t%s\n" % (contig_names[k >> 32], p1, p1, ref, INFO_FIELD, "\t".join(cols)))
    return "".join(out)


# ---- native writers (liblongsom_io.so, csrc/hostio/tsvwrite.cpp): same bytes as the format_* functions above, written straight
# to the file by a thread pool; the Python formatters stay as their test reference and for small in-memory uses ----------------
def _io():
    import ctypes as C
    from . import hostio
    lib = hostio.load()
    if not getattr(lib, "_tsv_ready", False):
        PP = C.POINTER(C.c_void_p)
        lib.lsio_tsv_last_error.restype = C.c_char_p
        lib.lsio_write_count_rows.restype = C.c_int
        lib.lsio_write_count_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        lib.lsio_write_merged_rows.restype = C.c_int
        lib.lsio_write_merged_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, PP, PP, PP, C.c_void_p, C.c_int32]
        lib.lsio_write_step1_rows.restype = C.c_int
        lib.lsio_write_step1_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_int64, PP, PP, C.c_void_p, C.c_int32,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.lsio_free_text.argtypes = [C.c_void_p]
        lib.lsio_scan_last_error.restype = C.c_char_p
        lib.lsio_scan_rows.restype = C.c_int
        lib.lsio_scan_rows.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p]
        lib.lsio_free_row_scan.argtypes = [C.c_void_p]
        lib.lsio_step3_last_error.restype = C.c_char_p
        lib.lsio_step3_rows.restype = C.c_int
        lib.lsio_step3_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.lsio_step3_column_kinds.restype = C.c_int
        lib.lsio_step3_column_kinds.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.lsio_gather_lines.restype = C.c_int
        lib.lsio_gather_lines.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
        lib._tsv_ready = True
    return lib


SCAN_ALT_DOT, SCAN_FILTER_DOT, SCAN_PAT_A, SCAN_PAT_B, SCAN_CT_MATCH, SCAN_SHORT, SCAN_BAD_POS, SCAN_UNKNOWN_CHROM = 1, 2, 4, 8, 16, 32, 64, 128


class RowScan:
    """what lsio_scan_rows (csrc/hostio/tsvscan.cpp) found in every row of a step-1 / step-2 table: off / len of the line in the text,
    key = tid << 32 | Start, the FILTER field's place inside the line, flags (SCAN_*); n_comment_lines = lines starting with '#'."""
    __slots__ = ("n_rows", "n_comment_lines", "off", "len", "key", "filt_off", "filt_len", "flags")


def scan_rows(text: bytes, contig_names, patterns_a: str = "", patterns_b: str = "", ct_col: int = -1, ct_value: str = "", threads: int = 0) -> RowScan:
    import ctypes as C

    class _S(C.Structure):
        _fields_ = [("n_rows", C.c_int64), ("n_comment_lines", C.c_int64), ("off", C.c_void_p), ("key", C.c_void_p), ("len", C.c_void_p),
                    ("filt_off", C.c_void_p), ("filt_len", C.c_void_p), ("flags", C.c_void_p)]
    lib = _io()
    st = _S()
    rc = lib.lsio_scan_rows(text, len(text), "\n".join(contig_names).encode(), len(contig_names), patterns_a.encode(), patterns_b.encode(), ct_col,
                            ct_value.encode() if ct_col >= 0 else None, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError("lsio_scan_rows: %s" % lib.lsio_scan_last_error().decode("utf-8", "replace"))
    try:
        r = RowScan()
        n = r.n_rows = int(st.n_rows)
        r.n_comment_lines = int(st.n_comment_lines)
        for name, dt in (("off", np.int64), ("key", np.int64), ("len", np.int32), ("filt_off", np.int32), ("filt_len", np.int32), ("flags", np.uint32)):
            a = np.empty(n, dt)
            if n:
                C.memmove(a.ctypes.data, getattr(st, name), n * a.itemsize)
            setattr(r, name, a)
        return r
    finally:
        lib.lsio_free_row_scan(C.byref(st))


STEP3_COLUMNS = ("#CHROM", "Start", "REF", "ALT", "FILTER", "Cell_types", "Dp", "Nc", "Bc", "Cc", "VAF", "MCF", "Cell_type_Filter", "Cancer", "Non-Cancer")


KIND_NA, KIND_INT, KIND_FLOAT, KIND_ODD, KIND_OTHER = 1, 2, 4, 8, 16


def column_kinds(text: bytes, n_cols: int) -> np.ndarray:
    """per column of a whole table (comment lines skipped): OR of KIND_* over every row's cell (lsio_step3_column_kinds) - what pandas'
    dtype inference over the WHOLE step-2 table sees, also in the rows step 3 drops before it parses anything"""
    lib = _io()
    out = np.zeros(int(n_cols), np.uint8)
    if lib.lsio_step3_column_kinds(text, len(text), int(n_cols), out.ctypes.data) != 0:
        raise RuntimeError(lib.lsio_step3_last_error().decode("utf-8", "replace"))
    return out


def kinds_dtype_sensitive(kinds: np.ndarray) -> bool:
    """a column pandas reads as numbers whose printed form depends on the dtype: an odd number, or integers beside missing / float cells"""
    k = np.asarray(kinds, np.uint8)
    num = (k & KIND_OTHER) == 0
    return bool(np.any(num & (((k & KIND_ODD) != 0) | (((k & KIND_INT) != 0) & ((k & (KIND_NA | KIND_FLOAT)) != 0)))))


def step3_rows(rows: bytes, cols, delta_vaf: float, delta_mcf: float, min_ac_reads: int, min_ac_cells: int, clust_dist: int, all_kinds=None, prefix: bytes = b"",
               skip: int = 0):
    """lsio_step3_rows (csrc/hostio/tsvstep3.cpp) over the surviving rows of a step-2 table whose header is `cols`: (rows of the
    unfiltered table, rows of the final table) as bytes, or None when the table is one for the pandas path.  all_kinds: column_kinds of
    the WHOLE table the rows were taken from (the dropped rows' cells decide pandas' dtypes too).  skip: the rows start at rows[skip] (a
    header in front of a gigabyte is not worth a copy of it)."""
    import ctypes as C
    lib = _io()
    rows = rows if isinstance(rows, bytes) else bytes(rows)
    idx = []
    for name in STEP3_COLUMNS:
        if name in cols:
            idx.append(cols.index(name))
        elif name == "Non-Cancer":
            idx.append(-1)
        else:
            return None
    col = (C.c_int32 * len(idx))(*idx)
    a = C.c_void_p(); na = C.c_int64(0); b = C.c_void_p(); nb = C.c_int64(0)
    kinds = None if all_kinds is None else np.ascontiguousarray(all_kinds, np.uint8)
    if kinds is not None and len(kinds) != len(cols):
        raise ValueError("all_kinds has %d entries for %d columns" % (len(kinds), len(cols)))
    rc = lib.lsio_step3_rows(C.cast(C.c_char_p(rows), C.c_void_p).value + int(skip), len(rows) - int(skip), len(cols), col, float(delta_vaf), float(delta_mcf), int(min_ac_reads), int(min_ac_cells), int(clust_dist),
                             None if kinds is None else kinds.ctypes.data, C.byref(a), C.byref(na), C.byref(b), C.byref(nb))
    if rc == 1:
        return None
    if rc != 0:
        raise RuntimeError(lib.lsio_step3_last_error().decode("utf-8", "replace"))
    try:
        return _take_bytes(a.value, na.value, prefix), _take_bytes(b.value, nb.value, prefix)       # (prefix: the tables' header, so that nobody concatenates a gigabyte afterwards)
    finally:
        lib.lsio_free_text(a); lib.lsio_free_text(b)


def gather_lines(text: bytes, off: np.ndarray, length: np.ndarray, blank_na: bool = False, threads: int = 0, prefix: bytes = b""):
    """(prefix + the lines text[off[i] : off[i] + length[i]] + b"\\n", concatenated; start of every line BEHIND the prefix, n + 1 entries).
    blank_na: fields other than a line's first that are exactly "NA" come out empty (lsio_gather_lines).  prefix: a table's header, so
    that nobody concatenates gigabytes to it afterwards."""
    import ctypes as C
    lib = _io()
    off = np.ascontiguousarray(off, np.int64); length = np.ascontiguousarray(length, np.int32)
    n = len(off)
    new_off = np.zeros(n + 1, np.int64)
    txt = C.c_void_p(); ln = C.c_int64(0)
    rc = lib.lsio_gather_lines(text, off.ctypes.data, length.ctypes.data, n, 1 if blank_na else 0, threads, C.byref(txt), C.byref(ln), new_off.ctypes.data)
    if rc != 0:
        raise RuntimeError("lsio_gather_lines: %s" % lib.lsio_scan_last_error().decode("utf-8", "replace"))
    try:
        return _take_bytes(txt.value, ln.value, prefix), new_off
    finally:
        lib.lsio_free_text(txt)


def write_bytes(path: str, data: bytes, threads: int = 0) -> None:
    """data -> path (replacing it), the copy into the page cache spread over the host's threads (lsio_write_bytes): the step-2 table of a
    10 M-read sample is 2.7 GB, and one thread's write() moves about 1 GB/s"""
    import ctypes as C
    lib = _io()
    lib.lsio_write_bytes.restype = C.c_int
    lib.lsio_write_bytes.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int32]
    if len(data) < (1 << 24):
        with open(path, "wb") as f:
            f.write(data)
        return
    _check(lib, lib.lsio_write_bytes(os.fsencode(path), data, len(data), threads), "lsio_write_bytes")


def _take_bytes(ptr, n: int, prefix: bytes = b"") -> bytes:
    """prefix + the n bytes at ptr as a bytes object (ctypes.string_at takes a C int: the kept rows of a 10 M-read step-1 table are
    2.7 GB).  Large ones are copied by the host's threads into a bytes object made for them (lsio_copy_bytes): one thread's copy of 2.7 GB
    into fresh pages is a second of wall, and a `header + rows` afterwards another."""
    import ctypes as C
    if not n:
        return bytes(prefix)
    if n < (1 << 24):
        return bytes(prefix) + bytes(memoryview((C.c_char * n).from_address(ptr)))
    lib = _io()
    api = C.pythonapi
    api.PyBytes_FromStringAndSize.restype = C.py_object
    api.PyBytes_FromStringAndSize.argtypes = [C.c_void_p, C.c_ssize_t]
    api.PyBytes_AsString.restype = C.c_void_p
    api.PyBytes_AsString.argtypes = [C.py_object]
    out = api.PyBytes_FromStringAndSize(None, len(prefix) + n)            # (uninitialised: filled below before anybody else sees it)
    at = api.PyBytes_AsString(out)
    if prefix:
        C.memmove(at, prefix, len(prefix))
    lib.lsio_copy_bytes.restype = C.c_int
    lib.lsio_copy_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.lsio_copy_bytes(at + len(prefix), ptr, n, 0)
    return out


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError("%s: %s" % (what, lib.lsio_tsv_last_error().decode("utf-8", "replace")))


def _per_ct_ptrs(per_ct):
    import ctypes as C
    ks = [np.ascontiguousarray(k, np.int64) for k, _, _ in per_ct]
    rs = [np.ascontiguousarray(r, np.uint8) for _, r, _ in per_ct]
    cs = [np.ascontiguousarray(c, np.uint32).reshape(-1, 42) for _, _, c in per_ct]
    arr = lambda xs: (C.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
    n = np.asarray([len(k) for k in ks], np.int64)
    return ks, rs, cs, arr(ks), arr(rs), arr(cs), n


def counts_header(sample_id, date_line=None) -> str:
    return "".join([date_line or file_date(), _CONCEPTS, "\t".join(["#CHROM", "POS", "REF", "INFO", str(sample_id)]) + "\n"])


def merged_header(celltype_names, date_line=None) -> str:
    return "".join([date_line or file_date(), _CONCEPTS, "\t".join(["#CHROM", "Start", "End", "REF", "INFO"] + list(celltype_names)) + "\n"])


def step1_header(header_lines: List[str], celltype_names) -> str:
    return "".join(list(header_lines) + [l + "\n" for l in STEP1_INFO_LINES] +
                   ["\t".join(["#CHROM", "Start", "End", "REF", "\t".join(STEP1_COLUMNS), "INFO"] + list(celltype_names)) + "\n"])


def write_counts_tsv(path, keys, refs, counts, contig_names, sample_id, date_line=None, threads: int = 0, header: bool = True) -> None:
    """format_counts_tsv straight to `path` (BaseCellCounter.py:300-308).  header=False: the rows only (a window's piece of the
    table, the analogue of the reference's <chrom>__<start>__<end>.BaseCellCounts.temp, :12-19)."""
    lib = _io()
    with open(path, "w") as f:
        f.write(counts_header(sample_id, date_line) if header else "")
    k = np.ascontiguousarray(keys, np.int64); r = np.ascontiguousarray(refs, np.uint8); c = np.ascontiguousarray(counts, np.uint32).reshape(-1, 42)
    _check(lib, lib.lsio_write_count_rows(os.fsencode(path), "\n".join(contig_names).encode(), len(contig_names), k.ctypes.data, r.ctypes.data, c.ctypes.data,
                                          len(k), threads), "lsio_write_count_rows")


def write_merged_tsv(path, per_ct, contig_names, celltype_names, date_line=None, threads: int = 0, header: bool = True) -> List[str]:
    """format_merged_tsv straight to `path`; returns the '##' header lines (step 1 copies them through)."""
    lib = _io()
    head = [date_line or file_date(), _CONCEPTS]
    with open(path, "w") as f:
        f.write(merged_header(celltype_names, head[0]) if header else "")
    ks, rs, cs, pk, pr, pc, n = _per_ct_ptrs(per_ct)
    _check(lib, lib.lsio_write_merged_rows(os.fsencode(path), "\n".join(contig_names).encode(), len(contig_names), len(per_ct), pk, pr, pc, n.ctypes.data, threads),
           "lsio_write_merged_rows")
    return [l + "\n" for l in "".join(head).split("\n") if l.startswith("##")]


def write_step1_tsv(path, calls, per_ct, contig_names, celltype_names, header_lines: List[str], threads: int = 0, header: bool = True,
                    as_bytes: bool = False, collect: bool = True):
    """format_step1_tsv straight to `path`.  Returns the SMALL text step 2 needs: the comment lines, the column header and the
    rows its awk filter keeps (ALT != "." and FILTER != ".", BaseCellCalling.step2.py:23); header=False: the file gets the rows only
    and the kept rows come back without the header; as_bytes: as bytes (what calling.step2_bytes takes), not str.  collect=False: the
    table is written and nothing comes back (the kept rows were taken earlier: step1_kept_rows).  path=None: nothing is written, only
    the kept rows are formatted."""
    import ctypes as C
    lib = _io()
    head = step1_header(header_lines, celltype_names) if header else ""
    if path is not None:
        with open(path, "w") as f:
            f.write(head)
    ks, rs, cs, pk, pr, pc, n = _per_ct_ptrs(per_ct)
    calls = np.ascontiguousarray(calls)
    txt = C.c_void_p(); ln = C.c_int64(0)
    _check(lib, lib.lsio_write_step1_rows(os.fsencode(path) if path is not None else None, "\n".join(contig_names).encode(), len(contig_names), len(per_ct),
                                          "\n".join(celltype_names).encode(), calls.ctypes.data, len(calls), pk, pc, n.ctypes.data, threads,
                                          C.byref(txt) if collect else None, C.byref(ln) if collect else None), "lsio_write_step1_rows")
    if not collect:
        return None
    try:
        rows = _take_bytes(txt.value, ln.value, head.encode() if as_bytes else b"")      # (the header in front from the start: no `head + 2.7 GB` afterwards)
    finally:
        lib.lsio_free_text(txt)
    return rows if as_bytes else head + rows.decode()


def step1_kept_rows(calls, per_ct, contig_names, celltype_names, header_lines: List[str], threads: int = 0, as_bytes: bool = True):
    """what write_step1_tsv returns - header + the rows step 2 keeps - without writing the table: only those rows are formatted (a third
    of C2's 23.9 M), so steps 2 and 3 start while the tables are still on their way to the disk"""
    return write_step1_tsv(None, calls, per_ct, contig_names, celltype_names, header_lines, threads=threads, as_bytes=as_bytes)


def _p(k: int) -> str:
    """text of Python round(p, 4) given k = round(p,4)*1e4 (str() of the rounded float)"""
    return repr(k / 10000.0)


def _ratio(a: int, b: int) -> str:
    return str(round(a / float(b), 4))


def format_step1_tsv(calls, per_ct, contig_names, celltype_names, header_lines: List[str]) -> str:
    """calls: structured array of lsg_call records for EVERY merged site; per_ct as in format_merged_tsv.
    header_lines: the '##' lines of the merged file (copied through, step1.py:34-35)."""
    out = list(header_lines)
    out += [l + "\n" for l in STEP1_INFO_LINES]
    out.append("\t".join(["#CHROM", "Start", "End", "REF", "\t".join(STEP1_COLUMNS), "INFO"] + list(celltype_names)) + "\n")
    n_ct = len(celltype_names)
    idx = [dict(zip(k.tolist(), range(len(k)))) for k, _, _ in per_ct]
    keys = calls["key"]
    perm = sort_like_reference(keys, contig_names)
    base = "ACTG"
    for i in perm.tolist():
        c = calls[i]
        k = int(c["key"])
        p1 = (k & 0xFFFFFFFF) + 1
        cols = []
        for ct in range(n_ct):
            j = idx[ct].get(k)
            cols.append("NA" if j is None else row_text(per_ct[ct][2][j]))
        up = bytes(c["up_ctx"]).split(b"\0")[0].decode()
        if not up:
            up, down = ".", "."
        else:
            down = bytes(c["down_ctx"]).split(b"\0")[0].decode()
        sf = int(c["site_filter"])
        rest = []
        for s_alt, s_tot, pk in ((c["sum_alts_bc"], c["sum_dp"], c["noise_p_bc"]), (c["sum_alts_cc"], c["sum_nc"], c["noise_p_cc"])):
            rest.append("%d;%d;%s" % (s_alt, s_tot, "1" if int(c["sum_alts_bc"]) == 0 else ("nan" if int(pk) == -2 else _p(int(pk)))))
        ctmin = str(int(c["cell_types_min"]))
        if sf & SF_CANDIDATE:
            alts, cts, dps, ncs, bcs, ccs, vafs, mcfs, bcps, ccps, flt = [], [], [], [], [], [], [], [], [], [], []
            for ct in range(n_ct):
                if not (int(c["has_cand"]) >> ct) & 1:
                    continue
                na = int(c["n_alt"][ct])
                row = per_ct[ct][2][idx[ct][k]]
                dp, nc = int(row[0]), int(row[1])
                a = [base[int(c["alt"][ct][q])] for q in range(na)]
                bc = [int(c["alt_bc"][ct][q]) for q in range(na)]
                cc = [int(c["alt_cc"][ct][q]) for q in range(na)]
                alts.append("|".join(a)); cts.append(celltype_names[ct]); dps.append(str(dp)); ncs.append(str(nc))
                bcs.append("|".join(map(str, bc))); ccs.append("|".join(map(str, cc)))
                vafs.append("|".join(_ratio(b, dp) for b in bc)); mcfs.append("|".join(_ratio(x, nc) for x in cc))
                bcps.append("|".join(_p(int(c["p_bc"][ct][q])) for q in range(na)))
                ccps.append("|".join(_p(int(c["p_cc"][ct][q])) for q in range(na)))
                flt.append(CT_FILTER_NAMES[int(c["ct_filter"][ct])])
            site = [name for bit, name in SITE_FILTER_NAMES if sf & bit]
            if site:
                FILTER = ",".join(site)
            else:
                FILTER = "PASS" if "PASS" in flt else ",".join(flt)
            info = [",".join(alts), FILTER, ",".join(cts), up, down, str(len(set(alts))), ",".join(dps), ",".join(ncs), ",".join(bcs),
                    ",".join(ccs), ",".join(vafs), ",".join(mcfs), ",".join(bcps), ",".join(ccps), ctmin, ctmin, rest[0], rest[1], ".",
                    ",".join(flt)]
        else:
            FILTER = "Noisy_site" if sf & 16 else "."
            info = [".", FILTER, ".", up, down, ".", ".", ".", ".", ".", ".", ".", ".", ".", ctmin, ctmin, rest[0], rest[1], ".", "."]
        out.append("%s\t%d\t%d\t%s\t%s\t%s\t%s\n" % (contig_names[k >> 32], p1, p1, chr(int(c["ref"])), "\t".join(info), INFO_FIELD, "\t".join(cols)))
    return "".join(out)


def _row_from_text(data: str) -> np.ndarray:
    dp, nc, cc, bc, bq, bcf, bcr = data.split("|")
    r = np.zeros(ROW_WORDS, np.uint32)
    r[0] = int(dp); r[1] = int(nc)
    for off, vec in ((2, cc), (10, bc), (18, bq), (26, bcf), (34, bcr)):
        v = [int(x) for x in vec.split(":")]
        r[off:off + len(v)] = v
    return r


def parse_merged_tsv(path, contig_names):
    """MergeBaseCellCounts TSV -> (celltype names, per_ct [(keys, refs, counts)], '##' header lines)."""
    tid_of = {n: i for i, n in enumerate(contig_names)}
    header, cts, acc = [], None, None
    with open(path) as f:
        for line in f:
            if line.startswith("##"):
                header.append(line); continue
            line = line.rstrip("\n")
            if line.startswith("#CHROM"):
                cts = line.split("\t")[5:]
                acc = [([], [], []) for _ in cts]
                continue
            if not line:
                continue
            el = line.split("\t")
            key = (tid_of[el[0]] << 32) | (int(el[1]) - 1)
            for ct, data in enumerate(el[5:]):
                if not data.startswith("NA"):
                    acc[ct][0].append(key); acc[ct][1].append(ord(el[3][0])); acc[ct][2].append(_row_from_text(data))
    per_ct = []
    for k, r, c in acc or []:
        k = np.asarray(k, np.int64); perm = np.argsort(k, kind="stable")
        c = np.stack(c) if c else np.zeros((0, ROW_WORDS), np.uint32)
        per_ct.append((k[perm], np.asarray(r, np.uint8)[perm], c[perm]))
    return cts or [], per_ct, header


def contigs_of_tsv(paths) -> List[str]:
    """chromosome names of count / merged TSVs in order of first appearance (when no FASTA is at hand)."""
    seen = {}
    for p in paths:
        with open(p) as f:
            for line in f:
                if line and not line.startswith("#"):
                    seen.setdefault(line.split("\t", 1)[0], None)
    return list(seen)


_FASTA_UPPER = bytes(range(256)).upper()


def read_fasta(path):
    """Whole FASTA -> (names, list of upper-cased uint8 arrays).  inFasta.fetch(...).upper(), BaseCellCounter.py:202-203."""
    with open(path, "rb") as f:
        data = f.read()
    names, seqs = [], []
    at = 0 if data.startswith(b">") else data.find(b"\n>") + 1          # (0 when there is no record at all: the loop below then ends at once)
    if not data.startswith(b">") and at == 0:
        return names, seqs
    while at < len(data):                                               # data[at] == '>'
        e = data.find(b"\n", at)
        e = len(data) if e < 0 else e
        names.append(data[at + 1:e].split()[0].decode())
        nxt = data.find(b"\n>", e)
        nxt = len(data) if nxt < 0 else nxt + 1
        seqs.append(np.frombuffer(data[e + 1:nxt].translate(_FASTA_UPPER, b" \t\r\n\v\f"), dtype=np.uint8).copy())      # (one pass: blanks out, letters up)
        at = nxt
    return names, seqs
