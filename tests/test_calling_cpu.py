"""CPU: step 3 (host side, pandas) against the reference's golden outputs."""
import os

import numpy as np
import pytest

from longsom_amd import calling, tsvio

G = os.path.join(os.path.dirname(__file__), "golden")


def rd(name):
    return open(os.path.join(G, name)).read()


@pytest.mark.parametrize("prefix,clust", [("sample", 10000), ("sample.dist150", 150)])
def test_step3_matches_reference_golden(prefix, clust):
    final, unfiltered = calling.step3(rd(prefix + ".calling.step2.tsv"), 0.05, 0.3, 3, 2, clust)
    assert unfiltered == rd(prefix + ".calling.step3.unfiltered.tsv")
    assert final == rd(prefix + ".calling.step3.tsv")


def test_step3_empty_input_gives_header_only():
    text = "\n".join(l for l in rd("sample.calling.step2.tsv").split("\n") if l.startswith("#")) + "\n"
    final, unfiltered = calling.step3(text, 0.05, 0.3, 3, 2, 10000)
    assert final == unfiltered and final.rstrip("\n").split("\n")[-1].endswith("STEP3FILTER\tINDEX")


def test_posset_reader(tmp_path):
    import gzip
    names = ["chr1", "chr10", "chr2", "chrM"]
    p = os.path.join(G, "calling.pon_SR.tsv")
    keys = calling.read_posset_keys(p, names)
    assert len(keys) > 10 and (keys[1:] > keys[:-1]).all()
    gz = tmp_path / "x.tsv.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(p, "rb").read())
    assert (calling.read_posset_keys(str(gz), names) == keys).all()
    assert len(calling.read_posset_keys(str(gz), names, reference_gz_compat=True)) == 0
    assert len(calling.read_posset_keys("", names)) == 0 and len(calling.read_posset_keys("/nonexistent", names)) == 0


class _NumpyProbe:
    """stands in for the device position sets of step 2 (Engine.load_posset / probe_posset): membership by numpy, so that the host
    half of step 2 — the row handling, the tags, the NA fields, the gnomAD lookup — is pinned to the reference's files without a GPU"""
    def __init__(self):
        self.sets = {}

    def load_posset(self, kind, keys):
        self.sets[kind] = np.asarray(keys, np.int64)

    def probe_posset(self, kind, q):
        return np.isin(np.asarray(q, np.int64), self.sets[kind]).astype(np.uint8)


def test_step2_host_half_matches_reference_golden():
    import json
    names, _ = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    ed, sr, lr = (calling.read_posset_keys(os.path.join(G, "calling.%s.tsv" % k), names) for k in ("editing", "pon_SR", "pon_LR"))
    af = json.load(open(os.path.join(G, "calling.gnomad_af.json")))
    s1 = rd("sample.calling.step1.tsv")
    assert calling.step2(s1, _NumpyProbe(), names, ed, sr, lr, 0, af, 0.01) == rd("sample.calling.step2.tsv")
    assert calling.step2(s1, _NumpyProbe(), names, ed, sr, calling.read_posset_keys("", names), 150, af, 0.01) == rd("sample.dist150.calling.step2.tsv")
    # without a gnomAD source most rows pass through verbatim: the same rows as with an empty table of allele frequencies
    class _Empty(dict):
        def __bool__(self): return True
    assert calling.step2(s1, _NumpyProbe(), names, ed, sr, lr, 0, None, 0.01) == calling.step2(s1, _NumpyProbe(), names, ed, sr, lr, 0, _Empty(), 0.01)


def test_step2_scanned_path_equals_row_path(monkeypatch):
    """step2_bytes without a gnomAD source goes over the native row scanner (csrc/hostio/tsvscan.cpp) and moves untagged rows as bytes;
    the row-by-row path (pinned to the reference's files above) must give the same table: goldens, both distances, a table with
    dropped rows / blank lines / no trailing newline, a replicated 40 k-row table, and the cases the scanner hands back"""
    names, _ = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    ed, sr, lr = (calling.read_posset_keys(os.path.join(G, "calling.%s.tsv" % k), names) for k in ("editing", "pon_SR", "pon_LR"))
    s1 = rd("sample.calling.step1.tsv")
    lines = s1.split("\n")
    comments = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    big, k = [], 0
    while len(big) < 40000:
        for l in body:
            f = l.split("\t"); f[1] = str(int(f[1]) + 7 * k); f[2] = f[1]; big.append("\t".join(f))
        k += 1
    texts = [s1, s1.rstrip("\n"), "\n".join(comments + body[:3]) + "\n\n" + "\n".join(body[3:]) + "\n", "\n".join(comments + big) + "\n",
             "\n".join(comments) + "\n", "\n".join(comments + body[:1]) + "\n", "\n".join(comments + body[:2]) + "\n"]
    for t in texts:
        for dist in (0, 150, 5):
            for sets in ((ed, sr, lr), (ed, sr, calling.read_posset_keys("", names))):
                monkeypatch.setenv("LONGSOM_STEP2_ROW_PATH", "1")
                want = calling.step2_bytes(t.encode(), _NumpyProbe(), names, *sets, dist)
                assert want.decode() == calling.step2(t, _NumpyProbe(), names, *sets, dist)
                monkeypatch.setenv("LONGSOM_STEP2_ROW_PATH", "0")
                assert calling._step2_scanned(t.encode(), _NumpyProbe(), names, *sets, dist) is not None
                assert calling.step2_bytes(t.encode(), _NumpyProbe(), names, *sets, dist) == want
    # handed back: a Start that is not a plain number, a comment line among the rows
    bad = "\n".join(comments + [body[0].replace("\t" + body[0].split("\t")[1] + "\t", "\t+5\t", 1)] + body[1:]) + "\n"
    assert calling._step2_scanned(bad.encode(), _NumpyProbe(), names, ed, sr, lr, 0) is None
    late = "\n".join(comments + body[:2] + ["#late"] + body[2:]) + "\n"
    assert calling._step2_scanned(late.encode(), _NumpyProbe(), names, ed, sr, lr, 0) is None
    assert calling.step2_bytes(late.encode(), _NumpyProbe(), names, ed, sr, lr, 0).decode() == calling.step2(late, _NumpyProbe(), names, ed, sr, lr, 0)


def test_step3_prefilter_equals_full_parse(monkeypatch):
    """step 3 parses only the rows its FILTER patterns let through; LONGSOM_STEP3_FULL_PARSE=1 parses every row like the reference:
    same files, on the goldens and on a 60 k-row table replicated from them (mixed single- and two-cell-type rows in every chunk)"""
    texts = [rd("sample.calling.step2.tsv"), rd("sample.dist150.calling.step2.tsv")]
    lines = texts[0].split("\n")
    comments = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    big, k = [], 0
    while len(big) < 60000:
        for l in body:
            f = l.split("\t"); f[1] = str(int(f[1]) + 1000 * k); f[2] = f[1]; big.append("\t".join(f))
        k += 1
    texts.append("\n".join(comments + big) + "\n")
    texts.append("\n".join(comments + [l for l in big if "Noisy_site" in l][:100]) + "\n")          # nothing survives
    for t in texts:
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "1")
        want = calling.step3(t, 0.05, 0.3, 3, 2, 10000)
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "0")
        assert calling.step3(t, 0.05, 0.3, 3, 2, 10000) == want                    # the native row scanner picks the survivors
        assert calling.step3(t.encode(), 0.05, 0.3, 3, 2, 10000) == want
        monkeypatch.setenv("LONGSOM_STEP3_ROW_PATH", "1")
        assert calling.step3(t, 0.05, 0.3, 3, 2, 10000) == want                    # the Python line loop does
        monkeypatch.setenv("LONGSOM_STEP3_ROW_PATH", "0")


def _step3_tables():
    texts = [rd("sample.calling.step2.tsv"), rd("sample.dist150.calling.step2.tsv")]
    lines = texts[0].split("\n")
    comments = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    big, k = [], 0
    while len(big) < 30000:
        for l in body:
            f = l.split("\t"); f[1] = str(int(f[1]) + 1000 * k); f[2] = f[1]; big.append("\t".join(f))
        k += 1
    texts.append("\n".join(comments + big) + "\n")
    # one cell type only: the Cancer rows without the Non-Cancer column — Dp, Nc, Bc, Cc, VAF, MCF are then NUMERIC columns for pandas
    cols = [l for l in comments if l.startswith("#CHROM")][0].split("\t")
    i_nc = cols.index("Non-Cancer")
    one = [l for l in comments if not l.startswith("#CHROM")] + ["\t".join(c for i, c in enumerate(cols) if i != i_nc)]
    for l in big:
        f = l.split("\t")
        if f[cols.index("Cell_types")] == "Cancer":
            one.append("\t".join(x for i, x in enumerate(f) if i != i_nc))
    texts.append("\n".join(one) + "\n")
    return texts, comments, body, cols


def test_step3_native_rows_equal_the_pandas_path(monkeypatch):
    """csrc/hostio/tsvstep3.cpp does step 3's row functions, drops, cluster filter and both tables natively; the pandas implementation
    (pinned to the reference's files by test_step3 above) must give the same bytes — on the goldens (which the native path handles: it is
    what test_step3 now pins), a 30 k-row table, a one-cell-type table whose count columns are numeric for pandas, three cluster distances"""
    texts, _, _, _ = _step3_tables()
    for t in texts:
        sv = calling._step3_survivors(t.encode(), 6)
        cols = [l for l in t.split("\n") if l.startswith("#CHROM")][0].split("\t")
        for clust in (10000, 150, 1):
            args = (0.05, 0.3, 3, 2, clust)
            assert tsvio.step3_rows(sv, cols, *args) is not None, "the native path hands this table back"
            monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "1")
            want = calling.step3(t, *args)
            monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "0")
            assert calling.step3(t, *args) == want
            assert calling.step3(t.encode(), *args) == want


def test_step3_over_survivors_and_the_whole_tables_kinds():
    """what the fused chain hands step 3 when the device printed the step-2 table: header + the rows its FILTER patterns and Cell_types
    test let through + the kinds of cell of every column over the WHOLE table (survivors_only: the rows are not looked for again, their
    cells not classified again) - the same two tables as step 3 over the whole text"""
    texts, _, _, _ = _step3_tables()
    for t in texts:
        tb = t.encode()
        head = b"".join(l + b"\n" for l in tb.split(b"\n") if l.startswith(b"#"))
        cols = [l for l in t.split("\n") if l.startswith("#CHROM")][0].split("\t")
        kinds = tsvio.column_kinds(tb, len(cols))
        part = head + calling._step3_survivors(tb, 6)
        for clust in (10000, 150):
            args = (0.05, 0.3, 3, 2, clust)
            want = calling.step3_bytes(tb, *args)
            assert calling.step3_bytes(part, *args, all_kinds=kinds, survivors_only=True) == want
            assert calling.step3_bytes(part, *args, all_kinds=kinds) == want
    # no survivors at all: header-only tables
    t = texts[0]
    head = "".join(l + "\n" for l in t.split("\n") if l.startswith("#")).encode()
    cols = [l for l in t.split("\n") if l.startswith("#CHROM")][0].split("\t")
    empty = calling.step3_bytes(head, 0.05, 0.3, 3, 2, 10000, all_kinds=np.zeros(len(cols), np.uint8), survivors_only=True)
    assert empty[0] == empty[1] and empty[0].endswith(b"STEP3FILTER\tINDEX\n")


def test_step3_native_hands_back_what_pandas_dtypes_could_change(monkeypatch):
    """an integer column with a missing value (pandas prints 12.0), a number that is not its own shortest repr, a '#', a quote, a row
    function that raises in Python: tsvstep3.cpp returns 'not mine' and the result (or the exception) is the pandas path's"""
    texts, comments, body, cols = _step3_tables()
    i_ct, i_f = cols.index("Cell_types"), cols.index("FILTER")
    live = [l for l in body if l.split("\t")[i_ct] != "Non-Cancer" and "PASS" in l.split("\t")[i_f]]
    assert len(live) >= 3

    def table(edit):
        rows = list(live)
        f = rows[1].split("\t"); edit(f); rows[1] = "\t".join(f)
        return "\n".join(comments + rows) + "\n"

    def setcol(name, value):
        def e(f): f[cols.index(name)] = value
        return e
    args = (0.05, 0.3, 3, 2, 10000)

    def outcome(t):
        try:
            return calling.step3(t, *args)
        except Exception as e:                          # (a table the reference's script dies on: the same death on both paths)
            return type(e)
    plain = table(lambda f: None)
    assert tsvio.step3_rows(calling._step3_survivors(plain.encode(), 6), cols, *args) is not None
    for name, value in (("N_ALT", ""), ("N_ALT", "01"), ("N_ALT", "1.50"), ("N_ALT", "1e3"), ("Up_context", "AC#GT"), ("Up_context", '"ACGT'), ("Cell_types_min_BC", "nan")):
        t = table(setcol(name, value))
        assert tsvio.step3_rows(calling._step3_survivors(t.encode(), 6), cols, *args) is None, (name, value)
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "1")
        want = outcome(t)
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "0")
        assert outcome(t) == want
    # a missing value in a column of strings prints as "" on both paths, and "nan" in one is a missing value for pandas
    for name, value in (("Up_context", ""), ("Up_context", "nan"), ("Fisher_p", "NULL")):
        t = table(setcol(name, value))
        assert tsvio.step3_rows(calling._step3_survivors(t.encode(), 6), cols, *args) is not None, (name, value)
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "1")
        want = calling.step3(t, *args)
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "0")
        assert calling.step3(t, *args) == want
    # an ALT the reference's BC_CC_filtering cannot index: both paths raise
    t = table(setcol("ALT", "N"))
    assert tsvio.step3_rows(calling._step3_survivors(t.encode(), 6), cols, *args) is None
    with pytest.raises(ValueError):
        calling.step3(t, *args)


def test_step3_native_differential_fuzz(monkeypatch):
    """random cells of random rows replaced by values that steer every branch of step 3 (missing values, one / two cell types in either
    order, multi-allelic ALTs, depths around the chrM thresholds, numbers that are not their own repr, all rows moved to chrM): whatever
    the native path does not hand back must come out as the pandas path prints it, and what kills the reference's script kills both"""
    import random
    texts, _, _, _ = _step3_tables()
    vals = ["", "NA", "nan", "0", "1", "12", "0.5", "1.0", "0.25", "PASS", "Multi-allelic", "A", "T", "A|C", "C,C", "Cancer", "Cancer,Non-Cancer", "Non-Cancer,Cancer", "7,3",
            "100,200", "99,100", "0.9,0.1", "chrM", "chr1", "Low-Significance", "PASS,Non-Significant", "Non-Significant,PASS", "LC_Upstream", "x",
            "5|5|1:2:3:4:0:0|5:6:7:8:0:0|1:1:1:1:0:0|1:1:1:1:0:0|1:1:1:1:0:0", "130|120|0:0:0:0:0:0|0:0:0:0:0:0", "0.00001", "1e-05", "-1", "00", "None"]
    args = (0.05, 0.3, 3, 2, 10000)

    def outcome(t):
        try:
            return calling.step3(t, *args)
        except Exception as e:
            return type(e).__name__
    handled = 0
    for seed in range(120):
        rng = random.Random(seed)
        monkeypatch.setenv("LONGSOM_STEP3_AHEAD_MIN", "0" if seed % 2 else str(1 << 26))      # (every other table finds its survivors on the thread big tables use)
        lines = texts[rng.choice([0, 1, 3])].split("\n")
        head = [l for l in lines if l.startswith("#")]
        rows = [l for l in lines if l and not l.startswith("#")]
        rows = rng.sample(rows, min(len(rows), 300))
        hdr = [l for l in head if l.startswith("#CHROM")][0].split("\t")
        for _ in range(rng.choice([0, 1, 1, 2, 5, 40])):
            i = rng.randrange(len(rows)); f = rows[i].split("\t")
            f[rng.randrange(len(f))] = rng.choice(vals); rows[i] = "\t".join(f)
        if rng.random() < 0.3:
            rows = ["\t".join(["chrM"] + l.split("\t")[1:]) for l in rows]
        t = "\n".join(head + rows) + "\n"
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "1")      # every row through pandas, as the reference reads the table: the ground truth
        want = outcome(t)
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "0")
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "1")          # pandas over the surviving rows, with the whole table's dtypes
        assert outcome(t) == want, "seed %d (pandas over the survivors)" % seed
        monkeypatch.setenv("LONGSOM_STEP3_PANDAS", "0")
        assert outcome(t) == want, "seed %d" % seed
        sv = calling._step3_survivors(t.encode(), hdr.index("Cell_types"))
        handled += bool(sv) and tsvio.step3_rows(sv, hdr, *args) is not None
    assert handled > 60                                  # (most of the tables are the native path's own)


def test_step3_dtypes_come_from_every_row_of_the_table(monkeypatch):
    """pandas infers a column's dtype over the WHOLE step-2 table: a missing or a float cell in a row step 3 DROPS (Non-Cancer, or dead by its
    FILTER) turns an integer column into float64 and its surviving cells into "12.0"; a string there turns it into a column of strings.
    The paths that parse only the survivors must print what the full parse prints."""
    texts, _, _, _ = _step3_tables()
    args = (0.05, 0.3, 3, 2, 10000)
    lines = texts[0].split("\n")
    head = [l for l in lines if l.startswith("#")]
    rows = [l.split("\t") for l in lines if l and not l.startswith("#")]
    hdr = [l for l in head if l.startswith("#CHROM")][0].split("\t")
    i_ct, i_f = hdr.index("Cell_types"), hdr.index("FILTER")
    victims = [r for r in rows if r[i_ct] == "Non-Cancer"][:2]
    assert len(victims) == 2
    n_diff = 0
    for col, cell in (("End", "NA"), ("End", "7.5"), ("Start", "x1"), ("N_ALT", ""), ("Dp", "3.0")):
        saved = [list(v) for v in victims]
        victims[0][hdr.index(col)] = cell
        t = "\n".join(head + ["\t".join(r) for r in rows]) + "\n"
        for v, sv_ in zip(victims, saved):
            v[:] = sv_
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "1")
        want = calling.step3(t, *args)
        monkeypatch.setenv("LONGSOM_STEP3_FULL_PARSE", "0")
        for pandas_only in ("1", "0"):
            monkeypatch.setenv("LONGSOM_STEP3_PANDAS", pandas_only)
            assert calling.step3(t, *args) == want, (col, cell, pandas_only)
        n_diff += want != calling.step3(texts[0], *args)
    assert n_diff >= 2                                    # (the dropped row's cell really changed what the survivors print)


def test_column_kinds_of_short_rows_and_of_a_comment_in_mid_line():
    """what pandas infers a dtype from: a row with fewer fields than columns gives the missing ones NA; read_csv(comment='#') cuts a line at a
    '#' anywhere, so the cells behind it are missing too and the cell it cuts is what is left of it"""
    K_NA, K_INT, K_FLOAT, K_OTHER = 1, 2, 4, 16
    t = b"#h\n1\t2\t3\n4\t5\n7\t8#x\t9.5\n"
    k = tsvio.column_kinds(t, 3)
    assert list(k) == [K_INT, K_INT, K_INT | K_NA]          # row 2 lacks column 3; row 3 is cut behind "8": its 9.5 is never seen
    k = tsvio.column_kinds(b"1\tx\n2\t3.5\n", 2)
    assert list(k) == [K_INT, K_OTHER | K_FLOAT] or list(k) == [K_INT, K_OTHER]


def test_step2_scanned_differential_fuzz(monkeypatch):
    """random subsets of the golden step-1 rows with cells replaced by missing values, dots, odd numbers and unknown contigs, blank and
    stray comment lines, with and without a final newline, four distances: the scanned path (or whatever it hands back) gives the
    row-by-row path's bytes, and what kills one kills the other"""
    import random
    names, _ = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    ed, sr, lr = (calling.read_posset_keys(os.path.join(G, "calling.%s.tsv" % k), names) for k in ("editing", "pon_SR", "pon_LR"))
    lines = rd("sample.calling.step1.tsv").split("\n")
    comments = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    vals = ["", "NA", ".", "PASS", "0", "12", "chrM", "chr1", "chrZ", "A", "A|C", "NA\tNA", "x", "-5", "+7", "1e3", "Multi-allelic", "00012"]

    def outcome(t, sets, dist):
        try:
            return calling.step2_bytes(t.encode(), _NumpyProbe(), names, *sets, dist)
        except Exception as e:
            return type(e).__name__
    for seed in range(150):
        rng = random.Random(seed)
        rows = rng.sample(body, min(len(body), rng.choice([0, 1, 2, 3, 50, 400])))
        if rng.random() < 0.7:
            rows.sort(key=body.index)
        for _ in range(rng.choice([0, 0, 1, 2, 8])):
            if not rows:
                break
            i = rng.randrange(len(rows)); f = rows[i].split("\t")
            f[rng.randrange(len(f))] = rng.choice(vals); rows[i] = "\t".join(f)
        if rng.random() < 0.1 and rows:
            rows.insert(rng.randrange(len(rows) + 1), "")
        if rng.random() < 0.05 and rows:
            rows.insert(rng.randrange(len(rows) + 1), "#stray")
        t = "\n".join(comments + rows) + ("\n" if rng.random() < 0.8 else "")
        dist = rng.choice([0, 0, 5, 150, 100000])
        sets = (ed, sr, lr) if rng.random() < 0.7 else (ed, sr, calling.read_posset_keys("", names))
        monkeypatch.setenv("LONGSOM_STEP2_ROW_PATH", "1")
        want = outcome(t, sets, dist)
        monkeypatch.setenv("LONGSOM_STEP2_ROW_PATH", "0")
        assert outcome(t, sets, dist) == want, "seed %d" % seed
