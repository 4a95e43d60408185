"""Steps 2 and 3 of the SNV call (host side; the data are the candidate rows only).

step2  <- variant_calling_step2 / GetExtraFilters   workflow/scripts/SNVCalling/BaseCellCalling.step2.py:14-235
          position-set membership (RNA editing, PoN_SR, PoN_LR) is probed on the GPU against sorted
          key arrays resident in HBM (lsg_load_posset / lsg_probe_posset).
step3  <- variant_calling_step3 + helpers           workflow/scripts/SNVCalling/BaseCellCalling.step3.py:8-316
          LongSom's final filters; the table goes through pandas exactly where the reference does, so
          the text (float formatting, empty cells) is the same.
"""
import gzip
import io
import os
import re
from typing import Dict, Optional, Sequence

import numpy as np
import pandas as pd

KIND_EDITING, KIND_PON_SR, KIND_PON_LR = 0, 1, 2


def read_posset_keys(path: Optional[str], contig_names: Sequence[str], reference_gz_compat: bool = False) -> np.ndarray:
    """Position-set file (col 0 chrom, col 1 1-based pos, '#' comments; build_dict, step2.py:197-221) ->
    sorted unique int64 keys (tid << 32) | pos1.  Unreadable / missing file -> empty (the reference's bare
    except).  reference_gz_compat=True reproduces SURVEY quirk Q1: a .gz path is opened as text by the
    reference, fails to decode and silently yields the empty set."""
    if not path:
        return np.zeros(0, np.int64)
    tid_of = {n: i for i, n in enumerate(contig_names)}
    keys = []
    try:
        if str(path).endswith(".gz"):
            if reference_gz_compat:
                return np.zeros(0, np.int64)
            fh = io.TextIOWrapper(gzip.open(path, "rb"))
        else:
            fh = open(path, "r")
        with fh:
            for line in fh:
                if line.startswith("#"):
                    continue
                el = line.split("\t")
                t = tid_of.get(el[0])
                p = int(el[1])
                if t is not None:
                    keys.append((t << 32) | p)
    except Exception:
        return np.zeros(0, np.int64)
    return np.unique(np.asarray(keys, np.int64))


class GnomadSqlite:
    """AF lookups in a gnomad_db sqlite (the database the reference's step 2 queries through the gnomad_db package,
    step2.py:100-108: gnomAD_DB(dir, gnomad_version="v4").get_info_from_df(df, "AF")): table gnomad_db keyed by
    (chrom without "chr", pos, ref, alt).  Read with the standard library; the package itself is not needed.  The database is
    not part of either repository, so this reader is unpinned; it exists so that the drop-in's gnomAD filter
    runs wherever the reference's does."""

    def __init__(self, path: str):
        import os
        import sqlite3
        f = os.path.join(path, "gnomad_db.sqlite3") if os.path.isdir(path) else path
        if not os.path.exists(f):
            raise FileNotFoundError(f)
        self._db = sqlite3.connect("file:%s?mode=ro" % f, uri=True)
        self._db.execute("SELECT AF FROM gnomad_db LIMIT 1")

    def get(self, key: str, default: float = 0.0) -> float:
        chrom, pos, ref, alt = key.split(":", 3)
        row = self._db.execute("SELECT AF FROM gnomad_db WHERE chrom = ? AND pos = ? AND ref = ? AND alt = ?",
                               (chrom[3:] if chrom.startswith("chr") else chrom, int(pos), ref, alt)).fetchone()
        return default if row is None or row[0] is None else float(row[0])

    def __bool__(self):
        return True


class GnomadUnusable(RuntimeError):
    """a gnomAD source was named but cannot be read (the reference stops in gnomAD_DB() then, BaseCellCalling.step2.py:100)"""


def open_gnomad(source: Optional[str], allow_missing: Optional[bool] = None):
    """--gnomAD_db / --gnomAD_json of the shims: a JSON {"chrom:pos:ref:alt": AF}, a gnomad_db directory or sqlite file, or
    nothing.  A source that is named but cannot be used STOPS the run, as the reference's gnomAD_DB() does — running step 2 without
    its germline filter is a different call set, not a degraded one.  allow_missing=True (--allow_missing_gnomad,
    Run.allow_missing_gnomad, LONGSOM_ALLOW_MISSING_GNOMAD=1): warn on stderr and run with the filter off."""
    import json
    import os
    import sys
    if not source:
        return None
    try:
        if str(source).endswith(".json"):
            return json.load(open(source))
        return GnomadSqlite(source)
    except Exception as e:                                      # noqa: BLE001 - every failure ends the same way
        if allow_missing is None:
            allow_missing = os.environ.get("LONGSOM_ALLOW_MISSING_GNOMAD", "0") == "1"
        msg = "gnomAD source %r cannot be used (%s: %s)" % (source, type(e).__name__, e)
        if not allow_missing:
            raise GnomadUnusable(msg + ": step 2 needs it for its germline filter; pass --allow_missing_gnomad (Run.allow_missing_gnomad: True, "
                                 "LONGSOM_ALLOW_MISSING_GNOMAD=1) to run without") from e
        sys.stderr.write("warning: " + msg + ": the gnomAD filter of step 2 is OFF, germline sites the reference would tag 'gnomAD' stay in the call set\n")
        return None


def _blank_na_fields(line: bytes) -> bytes:
    f = line.split(b"\t")
    return b"\t".join([f[0]] + [b"" if x == b"NA" else x for x in f[1:]])


def _step2_scanned(text: bytes, engine, contig_names, editing_keys, pon_sr_keys, pon_lr_keys, distance: int) -> Optional[bytes]:
    """step2 without a gnomAD source, over the row scanner (csrc/hostio/tsvscan.cpp): the keys of the probes come from the scan, the
    rows that get no tag (all but a handful) are moved as bytes, never split.  None = a table the scanner does not vouch for (short
    rows, a Start that is not a plain number, comment lines among the rows, unknown contigs with a distance filter): the caller then
    takes the row-by-row path below, which is also what tests compare this one with."""
    from . import tsvio
    if not text.endswith(b"\n"):
        text = text + b"\n"
    sc = tsvio.scan_rows(text, contig_names)
    if sc.n_rows and (sc.flags & (tsvio.SCAN_SHORT | tsvio.SCAN_BAD_POS)).any():
        return None
    head = text[:int(sc.off[0])] if sc.n_rows else text
    head_lines = [l for l in head.split(b"\n") if l]
    if len(head_lines) != sc.n_comment_lines or any(not l.startswith(b"#") for l in head_lines):
        return None
    comments = [l for l in head_lines if b"#CHROM" not in l]
    headers = [l for l in head_lines if b"#CHROM" in l]
    if not headers:
        return None
    kept = np.nonzero((sc.flags & (tsvio.SCAN_ALT_DOT | tsvio.SCAN_FILTER_DOT)) == 0)[0]         # awk filter, step2.py:23
    n = len(kept)
    q = np.ascontiguousarray(sc.key[kept])
    if distance > 0 and n and (sc.flags[kept] & tsvio.SCAN_UNKNOWN_CHROM).any():
        return None
    hits = []
    for kind, keys in ((KIND_EDITING, editing_keys), (KIND_PON_SR, pon_sr_keys), (KIND_PON_LR, pon_lr_keys)):
        engine.load_posset(kind, keys)
        hits.append(np.asarray(engine.probe_posset(kind, q) if len(keys) and n else np.zeros(n, np.uint8)).astype(bool))
    close = np.zeros(n, np.int64)
    if distance > 0 and n > 1:
        tid, pos = q >> 32, q & 0xFFFFFFFF

        def near(i, j):                                  # step2.py:59-92: same contig, another position, within `distance`
            return ((tid[i] == tid[j]) & (pos[i] != pos[j]) & (np.abs(pos[i] - pos[j]) <= distance)).astype(np.int64)
        if n < 3:
            close[0] += near(np.array([0]), np.array([1]))[0]; close[1] += near(np.array([1]), np.array([0]))[0]
        else:
            i = np.arange(n)
            close[1:] += near(i[1:], i[1:] - 1)
            close[:-1] += near(i[:-1], i[:-1] + 1)
            close[0] += near(np.array([0]), np.array([2]))[0]                                      # the first row's window is rows 0..2
    tagged = np.nonzero(hits[0] | hits[1] | hits[2] | (close > 0))[0]
    # (the table's head travels in front of the rows from the start, and the few tagged rows are spliced in between VIEWS of the others:
    # one copy of the table's 2.7 GB at C2 where `head + join(slices)` made three)
    hdr = b"\n".join(comments + [headers[-1]]) + b"\n"
    body, new_off = tsvio.gather_lines(text, sc.off[kept], sc.len[kept], blank_na=True, prefix=hdr)
    if len(tagged):
        mv, base = memoryview(body), len(hdr)
        pieces, at = [], 0
        for i in tagged.tolist():
            r = int(kept[i])
            line = text[int(sc.off[r]):int(sc.off[r]) + int(sc.len[r])]
            fo, fl = int(sc.filt_off[r]), int(sc.filt_len[r])
            F = line[fo:fo + fl]
            for on, t in ((hits[0][i], b"RNA_editing_db"), (close[i] > 0, b"Clustered"), (hits[1][i], b"PoN_SR"), (hits[2][i], b"PoN_LR")):
                if on:
                    F = t if F == b"PASS" else F + b"," + t
            pieces.append(mv[at:base + int(new_off[i])])
            pieces.append(_blank_na_fields(line[:fo] + F + line[fo + fl:]) + b"\n")
            at = base + int(new_off[i + 1])
        pieces.append(mv[at:])
        body = b"".join(pieces)
    return body


def step2_bytes(step1_text: bytes, engine, contig_names: Sequence[str], editing_keys, pon_sr_keys, pon_lr_keys, distance: int = 0,
                gnomad_af: Optional[Dict[str, float]] = None, gnomad_max: float = 0.01) -> bytes:
    """step2 on the bytes of the table (what the fused pipeline and the CLI hold): the scanned path when there is no gnomAD source
    (LONGSOM_STEP2_ROW_PATH=1 forces the row-by-row one), else step2() below."""
    have_af = isinstance(gnomad_af, GnomadSqlite) or bool(gnomad_af)
    if not have_af and os.environ.get("LONGSOM_STEP2_ROW_PATH", "0") != "1":
        out = _step2_scanned(step1_text, engine, contig_names, editing_keys, pon_sr_keys, pon_lr_keys, distance)
        if out is not None:
            return out
    return step2(step1_text.decode(), engine, contig_names, editing_keys, pon_sr_keys, pon_lr_keys, distance, gnomad_af, gnomad_max).encode()


def step2(step1_text: str, engine, contig_names: Sequence[str], editing_keys, pon_sr_keys, pon_lr_keys, distance: int = 0,
          gnomad_af: Optional[Dict[str, float]] = None, gnomad_max: float = 0.01) -> str:
    """Returns the text of <prefix>.calling.step2.tsv.  gnomad_af: {"chrom:pos:ref:alt": AF}; the gnomAD
    database itself is not part of this repository — absent entries count as AF 0."""
    if gnomad_af is None:
        gnomad_af = {}
    comments, header, rows, kept = [], None, [], []
    for line in step1_text.split("\n"):
        if line.startswith("#"):
            if "#CHROM" in line:
                header = line
            else:
                comments.append(line)
        elif line:
            el = line.split("\t", 6)                    # CHROM Start End REF ALT FILTER | the rest stays one string
            if el[4] != "." and el[5] != ".":           # awk filter, step2.py:23
                rows.append(el); kept.append(line)
    tid_of = {n: i for i, n in enumerate(contig_names)}
    pos = [int(el[1]) for el in rows]
    q = np.asarray([(tid_of.get(el[0], 0x7FFFFFFF) << 32) | p for el, p in zip(rows, pos)], np.int64)
    hits = []
    for kind, keys in ((KIND_EDITING, editing_keys), (KIND_PON_SR, pon_sr_keys), (KIND_PON_LR, pon_lr_keys)):
        engine.load_posset(kind, keys)
        hits.append((engine.probe_posset(kind, q) if len(keys) and len(q) else np.zeros(len(q), np.uint8)).tolist())
    n = len(rows)
    # the 3-row window of step2.py:59-92 (rows 0..2 for the first row); a neighbour at the SAME position never counts, so
    # distance 0 (LongSom's setting) tags nothing
    close = [0] * n
    if distance > 0:
        for i in range(n):
            lo, hi = (0, n) if n < 3 else ((0, 3) if i == 0 else (i - 1, min(n, i + 2)))
            chrom, p = rows[i][0], pos[i]
            c = 0
            for j in range(lo, hi):
                if rows[j][0] == chrom and pos[j] != p and abs(pos[j] - p) <= distance:
                    c += 1
            close[i] = c
    have_af = isinstance(gnomad_af, GnomadSqlite) or bool(gnomad_af)
    h_ed, h_sr, h_lr = hits
    # a row that gets no tag leaves as it came; the tagged ones (a handful among the hundreds of thousands of candidate rows of a real
    # sample; every row when a gnomAD source is given) get their FILTER rebuilt
    out = list(kept)
    if have_af:
        todo = range(n)
    else:
        tagged = np.asarray(h_ed, bool) | np.asarray(h_sr, bool) | np.asarray(h_lr, bool) | (np.asarray(close) > 0)
        todo = np.nonzero(tagged)[0].tolist()
    for i in todo:
        el = rows[i]
        F = el[5]
        tags = []
        if h_ed[i]: tags.append("RNA_editing_db")
        if close[i] > 0: tags.append("Clustered")
        if h_sr[i]: tags.append("PoN_SR")
        if h_lr[i]: tags.append("PoN_LR")
        if have_af:
            af = gnomad_af.get("%s:%s:%s:%s" % (el[0], el[1], el[3], el[4]), 0.0)
            if af == af and af >= gnomad_max:
                tags.append("gnomAD")
        if not tags:
            continue
        for t in tags:
            F = t if F == "PASS" else F + "," + t
        out[i] = "\t".join(el[:5] + [F] + el[6:])
    # pandas writes a missing field (the reference's read_csv takes a whole field "NA" for NaN, step2.py:96) as "": done on the whole
    # text at once — the first six fields of a row (CHROM Start End REF ALT FILTER) are never "NA", so only the per-cell-type columns
    # and their kin are touched, as before
    body = "\n".join(out)
    if "\tNA" in body:
        body = (body + "\n").replace("\tNA\n", "\t\n")
        while "\tNA\t" in body:                          # ("\tNA\tNA\t": the second one only matches once the first is gone)
            body = body.replace("\tNA\t", "\t\t")
        body = body[:-1]
    return "\n".join(comments + [header]) + "\n" + (body + "\n" if out else "")


_NA_FIELD = re.compile(r"(?:(?<=\t)|^)NA(?=\t|$)")     # a whole field equal to NA


# ---- step 3 ---------------------------------------------------------------------------------------
_ACTG = "ACTG"
FINAL_FILTER_LINE = "##INFO=FINAL_FILTER,Description=Final ilter status, including chrM contaminants and clustered sites\n"


def _tag(cur: str, t: str) -> str:
    return t if cur == "PASS" else cur + "," + t


def _cancer_index(ctypes):
    return (0, 1) if ctypes[0] == "Cancer" else (1, 0)


def _multiallelic(row):
    """MultiAllelic_filtering (step3.py:163-231): at multi-allelic sites keep the dominant A/C/T/G alt of the
    Cancer column, recompute its counts, tag 'Multi-Allelic' unless the runner-up is < 5 % of it."""
    ALT, FILTER, CT = row["ALT"], row["FILTER"], row["Cell_types"]
    if not ("Multi-allelic" in FILTER or "|" in ALT):
        return ALT, FILTER, CT, row["Bc"], row["Cc"], row["VAF"], row["MCF"], "PASS"
    i_ref = _ACTG.index(row["REF"])
    ctypes = CT.split(",")
    cinfo = row["Cancer"].split("|")
    bcs = [int(x) for x in cinfo[3].split(":")[:4]]
    bcs[i_ref] = 0
    top = int(np.argmax(bcs)); mx = bcs[top]
    bcs[top] = 0
    mx2 = max(bcs)
    step3 = "PASS" if mx2 / mx < 0.05 else "Multi-Allelic"
    alt = _ACTG[top]
    bc_c = int(cinfo[3].split(":")[top]); cc_c = int(cinfo[2].split(":")[top])
    if len(ctypes) > 1:
        i_c, i_n = _cancer_index(ctypes)
        ninfo = row["Non-Cancer"].split("|")
        bc_n = int(ninfo[3].split(":")[top]); cc_n = int(ninfo[2].split(":")[top])
        dps, ncs = row["Dp"].split(","), row["Nc"].split(",")
        vaf_c, mcf_c = round(bc_c / int(dps[i_c]), 4), round(cc_c / int(ncs[i_c]), 4)
        vaf_n, mcf_n = round(bc_n / int(dps[i_n]), 4), round(cc_n / int(ncs[i_n]), 4)
        # the reference writes the pair as (Non-Cancer, Cancer) whatever the order of Cell_types (:196-200)
        return (",".join([alt, alt]), FILTER, CT, "%d,%d" % (bc_n, bc_c), "%d,%d" % (cc_n, cc_c), "%s,%s" % (vaf_n, vaf_c),
                "%s,%s" % (mcf_n, mcf_c), step3)
    FILTER = FILTER.replace("Multi-allelic,", "").replace(",Multi-allelic", "").replace("Multi-allelic", "")
    return alt, FILTER, CT, bc_c, cc_c, round(bc_c / int(row["Dp"]), 4), round(cc_c / int(row["Nc"]), 4), step3


def _chrm(row, dvaf_min, dmcf_min):
    """chrM_filtering (step3.py:101-161)"""
    s3 = row["STEP3FILTER"]
    ctypes = row["Cell_types"].split(",")
    if len(ctypes) > 1:
        i_c, i_n = _cancer_index(ctypes)
        d1, d2 = str(row["Dp"]).split(",")
        if int(d1) < 100 or int(d2) < 100:
            return _tag(s3, "LowDepth")
        v = [float(x) for x in str(row["VAF"]).split(",")]; m = [float(x) for x in str(row["MCF"]).split(",")]
        if v[i_c] - v[i_n] < dvaf_min:
            return _tag(s3, "LowDeltaVAF")
        if m[i_c] - m[i_n] < dmcf_min:
            return _tag(s3, "LowDeltaMCF")
        return s3
    if int(row["Dp"]) < 100:
        return _tag(s3, "LowDepth")
    if float(row["VAF"]) < 0.05:
        return _tag(s3, "LowVAF")
    if float(row["MCF"]) < 0.05:
        return _tag(s3, "LowMCF")
    return s3


def _bc_cc(row, min_ac_reads, min_ac_cells):
    """BC_CC_filtering (step3.py:233-251): alt reads / cells in the Cancer column; a missing column (NaN after
    the pandas round trip) -> NoCov."""
    s3 = row["STEP3FILTER"]
    i_alt = _ACTG.index(row["ALT"][0])
    cancer = row["Cancer"]
    if not isinstance(cancer, str):
        return _tag(s3, "NoCov")
    info = cancer.split("|")
    if int(info[3].split(":")[i_alt]) < min_ac_reads or int(info[2].split(":")[i_alt]) < min_ac_cells:
        return _tag(s3, "LowDepth")
    return s3


def _betabin(row):
    """BetaBino_filtering (step3.py:254-280)"""
    s3 = row["STEP3FILTER"]
    ctypes = row["Cell_types"].split(",")
    flt = row["Cell_type_Filter"]
    weak = ("Non-Significant", "Low-Significance")
    if len(ctypes) == 1:
        return _tag(s3, "CancerNonSig") if flt in weak else s3
    i_c, i_n = _cancer_index(ctypes)
    f = flt.split(",")
    if f[i_c] in weak:
        return _tag(s3, "CancerNonSig")
    if f[i_n] in ("PASS", "Low-Significance"):
        return _tag(s3, "NonCancerSig")
    return s3


def _records(df, cols):
    """rows of df as dicts over the columns a filter reads (row-wise apply at a fraction of its cost)"""
    cols = [c for c in cols if c in df.columns]
    arrays = [df[c].tolist() for c in cols]
    return [dict(zip(cols, vals)) for vals in zip(*arrays)]


_DEAD_M = "Min|LR|gnomAD|LC|RNA"                       # step3.py:49-52, chrM rows
_DEAD_O = "Min_cell_types|Noisy_site|LC_Upstream|LC_Downstream|RNA_editing_db|PoN|Cell_type_noise|gnomAD"      # step3.py:60-84, the others


def _step3_survivors(text: bytes, i_ct: int) -> Optional[bytes]:
    """the rows of a step-2 table that step 3's FILTER patterns and its Cell_types test let through, found by the row scanner
    (csrc/hostio/tsvscan.cpp); None = rows the scanner does not vouch for (the caller then splits the lines itself)"""
    from . import tsvio
    if not 5 <= i_ct <= 7:
        return None
    if not text.endswith(b"\n"):
        text = text + b"\n"
    sc = tsvio.scan_rows(text, ["chrM"], _DEAD_M, _DEAD_O, i_ct, "Non-Cancer")
    fl = sc.flags
    is_m = (fl & tsvio.SCAN_UNKNOWN_CHROM) == 0
    dead = np.where(is_m, fl & tsvio.SCAN_PAT_A, fl & tsvio.SCAN_PAT_B) != 0
    keep = np.nonzero(((fl & (tsvio.SCAN_SHORT | tsvio.SCAN_CT_MATCH)) == 0) & ~dead)[0]
    return tsvio.gather_lines(text, sc.off[keep], sc.len[keep])[0]


class _SurvivorsAhead:
    """_step3_survivors of a table on a thread of its own, started before the caller knows whether it will want them"""

    def __init__(self, text: bytes, cols):
        import threading
        self.i_ct = cols.index("Cell_types") if cols and "Cell_types" in cols else 6
        self.out, self.err = None, None

        def work():
            try:
                self.out = _step3_survivors(text, self.i_ct)
            except BaseException as e:                        # noqa: BLE001 - raised again in result()
                self.err = e
        self.th = threading.Thread(target=work, daemon=True)
        self.th.start()

    def result(self, i_ct: int):
        self.th.join()
        if self.err is not None:
            raise self.err
        assert i_ct == self.i_ct
        return self.out


def step3_bytes(step2_text, delta_vaf: float, delta_mcf: float, min_ac_reads: int, min_ac_cells: int, clust_dist: int, all_kinds=None, full_text=None,
                survivors_only: bool = False):
    """step3 with both tables as bytes (what the fused pipeline writes: at C2's size the unfiltered table is 0.8 GB, not worth a decode
    and an encode).  all_kinds / full_text / survivors_only: see step3."""
    final, unfiltered = step3(step2_text, delta_vaf, delta_mcf, min_ac_reads, min_ac_cells, clust_dist, _as_bytes=True, all_kinds=all_kinds, full_text=full_text,
                              survivors_only=survivors_only)
    return (final if isinstance(final, bytes) else final.encode()), (unfiltered if isinstance(unfiltered, bytes) else unfiltered.encode())


def step3(step2_text, delta_vaf: float, delta_mcf: float, min_ac_reads: int, min_ac_cells: int, clust_dist: int, _as_bytes: bool = False,
          all_kinds=None, full_text=None, survivors_only: bool = False):
    """Returns (text of .calling.step3.tsv, text of .calling.step3.unfiltered.tsv).  step2_text: str or bytes.
    all_kinds: tsvio.column_kinds of the WHOLE step-2 table when step2_text holds only a part of its rows (the survivors the ranks of a
    sharded run send to rank 0); full_text: a callable that returns the whole table then - called only when that table's printed form
    depends on pandas' dtypes (a table this package did not write), to be parsed whole as the reference does.  survivors_only (with
    all_kinds): the rows of step2_text ARE the survivors of step 3's FILTER patterns and Cell_types test already (Engine.step2_summary
    picked them on the device by the same rules), they are not looked for again."""
    as_bytes = isinstance(step2_text, (bytes, bytearray, memoryview))
    comments, cols = [], None
    at = 0
    nl, hash_, tag = (b"\n", b"#", b"#CHROM") if as_bytes else ("\n", "#", "#CHROM")
    while at < len(step2_text):                           # the comment lines at the top of the table
        e = step2_text.find(nl, at)
        e = len(step2_text) if e < 0 else e
        line = step2_text[at:e]
        if not line.startswith(hash_):
            break
        line = line.decode() if as_bytes else line
        if "#CHROM" in line:
            cols = line.split("\t")
        else:
            comments.append(line + "\n")
        at = e + 1
    head = "".join(comments) + FINAL_FILTER_LINE
    # Only the rows that survive step 3's FILTER patterns (below) are parsed: a numeric field of this table is the shortest repr of
    # its value (Python's str() in step 1, pandas' in the reference's step 2), which pandas prints back unchanged - PROVIDED the column's
    # dtype is what the surviving rows alone would give it.  pandas infers dtypes over the whole file, so the kinds of cell of EVERY row
    # are collected first (tsvio.column_kinds: one native pass over the text): a column of integers that holds a missing or a float cell
    # in some dropped row would print "12.0", and such a table - none this package writes - is parsed whole, as the reference does.
    # LONGSOM_STEP3_FULL_PARSE=1 parses every row in any case (tests compare the two).
    full_parse = os.environ.get("LONGSOM_STEP3_FULL_PARSE", "0") == "1"
    dtypes = None
    if not full_parse and cols:
        from . import tsvio as _tsvio
        early_survivors = None
        if all_kinds is None:
            # (the survivors of the table - what the common case parses next - are found beside this pass, on a thread of their own: both are
            # native passes over the same gigabytes that leave the interpreter alone)
            tb = step2_text if as_bytes else step2_text.encode()
            early_survivors = _SurvivorsAhead(tb, cols) if len(tb) > int(os.environ.get("LONGSOM_STEP3_AHEAD_MIN", str(1 << 26))) and os.environ.get("LONGSOM_STEP3_ROW_PATH", "0") != "1" else None
            all_kinds = _tsvio.column_kinds(tb, len(cols))
        k = np.asarray(all_kinds, np.uint8)
        numeric = (k & _tsvio.KIND_OTHER) == 0
        if np.any(numeric & ((k & _tsvio.KIND_ODD) != 0)):
            # a number pandas reads but would not print back as it stands, somewhere in a numeric column: parse the whole table
            full_parse = True
            if full_text is not None:
                step2_text = full_text()
                as_bytes = isinstance(step2_text, (bytes, bytearray, memoryview))
        else:
            # the dtypes pandas infers over the WHOLE table, for a parse of the surviving rows alone: a column with a string anywhere is a
            # column of strings, one with a float or a missing cell anywhere is float64 (its integers print "12.0"), the rest int64
            dtypes = {cols[c]: (str if not numeric[c] else "int64" if (k[c] & (_tsvio.KIND_NA | _tsvio.KIND_FLOAT)) == 0 and (k[c] & _tsvio.KIND_INT) else "float64")
                      for c in range(len(cols))}
    if not full_parse:
        i_ct = cols.index("Cell_types") if cols and "Cell_types" in cols else 6
        survivors, skip = None, 0
        if survivors_only and as_bytes and all_kinds is not None and os.environ.get("LONGSOM_STEP3_RESCAN", "0") != "1":
            survivors, skip = step2_text, at                  # (`at`: the end of the comment lines, found above)
        elif os.environ.get("LONGSOM_STEP3_ROW_PATH", "0") != "1":
            survivors = early_survivors.result(i_ct) if cols and early_survivors is not None else _step3_survivors(step2_text if as_bytes else step2_text.encode(), i_ct)
        if survivors is None:
            dead_m, dead_o = re.compile(_DEAD_M), re.compile(_DEAD_O)
            keep_lines = []
            for line in (step2_text.decode() if as_bytes else step2_text).split("\n"):
                if not line or line.startswith("#"):
                    continue
                el = line.split("\t", i_ct + 1)
                if len(el) <= i_ct or el[i_ct] == "Non-Cancer":
                    continue
                if (dead_m if el[0] == "chrM" else dead_o).search(el[5]) is None:
                    keep_lines.append(line)
            survivors = ("\n".join(keep_lines) + "\n").encode() if keep_lines else b""
        if len(survivors) <= skip:
            empty = head + "\t".join(cols + ["STEP3FILTER", "INDEX"]) + "\n"
            return empty, empty
        step2_text = survivors
        as_bytes = True
        if cols and os.environ.get("LONGSOM_STEP3_PANDAS", "0") != "1":
            # the row functions, the drops, the cluster filter and the two tables natively (csrc/hostio/tsvstep3.cpp); None = a table whose
            # printed form could depend on pandas' dtypes, or on which a row function raises: the pandas path below decides
            from . import tsvio
            header = head + "\t".join(cols + ["STEP3FILTER", "INDEX"]) + "\n"
            done = tsvio.step3_rows(survivors, cols, delta_vaf, delta_mcf, min_ac_reads, min_ac_cells, clust_dist, all_kinds=all_kinds, prefix=header.encode(), skip=skip)
            if done is not None:
                if _as_bytes:
                    return done[1], done[0]
                return done[1].decode(), done[0].decode()
    df = pd.read_csv(io.BytesIO(step2_text) if as_bytes else io.StringIO(step2_text), sep="\t", comment="#", names=cols,
                     dtype=None if full_parse else dtypes)
    df = df[df["Cell_types"] != "Non-Cancer"]
    out_cols = cols + ["STEP3FILTER", "INDEX"]
    if len(df) == 0:
        # the reference crashes on an empty frame under pandas 2; emit header-only files instead
        empty = head + "\t".join(out_cols) + "\n"
        return empty, empty
    # Every filter below is row-local except the final cluster test, and the rows dropped by the FILTER patterns (step3.py:49-84: chrM
    # rows with Min|LR|gnomAD|LC|RNA; other rows with Min_cell_types, Noisy_site, LC_*, RNA_editing_db, PoN, Cell_type_noise, gnomAD) are
    # dropped whatever the row-wise functions did to them before — MultiAllelic_filtering only ever REMOVES "Multi-allelic" from FILTER,
    # which none of the patterns can match — so they are dropped FIRST and the Python row functions see the survivors only (the
    # reference runs them over every candidate row: 7 of the 7.7 s of step 3 on a 780 k-row step-2 table).
    is_m = (df["#CHROM"] == "chrM").to_numpy()
    flt = df["FILTER"].astype(str)
    dead_m = flt.str.contains("Min|LR|gnomAD|LC|RNA", regex=True).to_numpy()
    dead_o = flt.str.contains("Min_cell_types|Noisy_site|LC_Upstream|LC_Downstream|RNA_editing_db|PoN|Cell_type_noise|gnomAD", regex=True).to_numpy()
    df = df[np.where(is_m, ~dead_m, ~dead_o)]
    if len(df) == 0:
        empty = head + "\t".join(out_cols) + "\n"
        return empty, empty
    # MultiAllelic_filtering touches only rows flagged Multi-allelic or with several alts; every other row keeps its values and
    # gets STEP3FILTER = PASS (the per-row function's early return), so only the touched rows go through Python
    upd = ["ALT", "FILTER", "Cell_types", "Bc", "Cc", "VAF", "MCF", "STEP3FILTER"]
    df["STEP3FILTER"] = "PASS"
    multi = (df["FILTER"].astype(str).str.contains("Multi-allelic", regex=False) | df["ALT"].astype(str).str.contains("|", regex=False)).to_numpy()
    if multi.any():
        sub = df[multi]
        res = [_multiallelic(r) for r in _records(sub, ("ALT", "FILTER", "Cell_types", "Bc", "Cc", "VAF", "MCF", "REF", "Dp", "Nc", "Cancer", "Non-Cancer"))]
        newvals = pd.DataFrame(res, columns=upd, index=sub.index)
        for c in upd:
            col = df[c].astype(object)
            col[multi] = newvals[c].astype(object)
            df[c] = col
    df["INDEX"] = df["#CHROM"].astype(str) + ":" + df["Start"].astype(str) + ":" + df["ALT"].str.split(",", n=1, expand=True)[0]
    chrm = df[df["#CHROM"] == "chrM"].copy()
    df = df[df["#CHROM"] != "chrM"]
    chrm = chrm[~chrm["FILTER"].str.contains("Min|LR|gnomAD|LC|RNA", regex=True)]
    if len(chrm) > 0:
        chrm["STEP3FILTER"] = [_chrm(r, delta_vaf, delta_mcf) for r in _records(chrm, ("STEP3FILTER", "Cell_types", "Dp", "VAF", "MCF"))]
    df = df[~df["FILTER"].str.contains("Min_cell_types")]
    if len(df) > 0:
        df["STEP3FILTER"] = [_bc_cc(r, min_ac_reads, min_ac_cells) for r in _records(df, ("STEP3FILTER", "ALT", "Cancer"))]
        df["STEP3FILTER"] = [_betabin(r) for r in _records(df, ("STEP3FILTER", "Cell_types", "Cell_type_Filter"))]
    for pat in ("Noisy_site", "LC_Upstream|LC_Downstream", "RNA_editing_db", "PoN", "Cell_type_noise", "gnomAD"):
        df = df[~df["FILTER"].str.contains(pat, regex=True)]
    df = pd.concat([df, chrm])
    # 10 kb cluster filter among PASS rows, neighbours in STRING-sorted (chr, pos) order (step3.py:283-306)
    idx = [tuple(i.split(":")) for i in df[df["STEP3FILTER"] == "PASS"]["INDEX"]]
    idx.sort(key=lambda x: (x[0], x[1]))
    trash = set()
    for (c1, p1, b1), (c2, p2, b2) in zip(idx, idx[1:]):
        if c1 == c2 and c1 != "chrM" and abs(int(p1) - int(p2)) < clust_dist:
            trash.add(":".join([c1, p1, b1])); trash.add(":".join([c2, p2, b2]))
    tag = "Clust_dist_%s" % clust_dist
    if len(df) > 0:
        df["STEP3FILTER"] = [(_tag(f, tag) if i in trash else f) for i, f in zip(df["INDEX"], df["STEP3FILTER"])]
    unfiltered = head + df.to_csv(sep="\t", index=False)
    keep = df[~df["STEP3FILTER"].str.contains("dist", regex=True)] if len(df) else df
    keep = keep[keep["STEP3FILTER"] == "PASS"]
    return head + keep.to_csv(sep="\t", index=False), unfiltered
