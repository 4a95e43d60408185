"""GPU: the tables' text printed on the device (csrc/tables.hip: lsg_format_table) is byte for byte what the host writers print from the
fetched rows and call records (csrc/hostio/tsvwrite.cpp, themselves pinned to the Python formatters and through them to the reference's
golden tables in test_tsvwrite_cpu.py / test_tsv_cpu.py)."""
import os

import numpy as np
import pytest

from longsom_amd import tsvio
from longsom_amd._lib import CallParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def host_tables(tmp, per_ct, calls, names, ct_names):
    """the rows of every table as the host writers print them: {table id: bytes}"""
    out = {}
    for ct, rows in enumerate(per_ct):
        p = os.path.join(tmp, "c%d.tsv" % ct)
        tsvio.write_counts_tsv(p, *rows, names, "S", header=False)
        out[ct] = open(p, "rb").read()
    p = os.path.join(tmp, "m.tsv")
    tsvio.write_merged_tsv(p, per_ct, names, ct_names, header=False)
    out[4] = open(p, "rb").read()
    p = os.path.join(tmp, "s1.tsv")
    out[6] = tsvio.write_step1_tsv(p, calls, per_ct, names, ct_names, [], header=False, as_bytes=True)
    out[5] = open(p, "rb").read()
    return out


def step2_on_device_equals_host(engine, names, ct_names, kept_rows: bytes, possets):
    """table TABLE_STEP2 and lsg_step2_summary against calling._step2_scanned / tsvio.column_kinds / calling._step3_survivors over the text
    of the kept rows the host holds"""
    from longsom_amd import calling
    head = tsvio.step1_header(["##fileDate=01/01/2000\n"], ct_names).encode()
    want2 = calling._step2_scanned(head + kept_rows, engine, names, possets[0], possets[1], possets[2], 0)      # (loads the position sets)
    assert want2 is not None
    cols = head.decode().split("\n")[-2].split("\t")
    hdr = b"".join(l + b"\n" for l in head.split(b"\n") if l.startswith(b"#"))
    assert want2.startswith(hdr)
    n = engine.format_table(engine.TABLE_STEP2)
    got2 = engine.table_bytes(engine.TABLE_STEP2, n)
    assert got2 == want2[len(hdr):], "step-2 rows"
    kinds, n_surv = engine.step2_summary(len(cols))
    want_kinds = tsvio.column_kinds(want2, len(cols))
    other = (want_kinds & tsvio.KIND_OTHER) != 0
    assert ((kinds & tsvio.KIND_OTHER) != 0).tolist() == other.tolist()
    assert kinds[~other].tolist() == want_kinds[~other].tolist(), (kinds, want_kinds)       # (a column of strings stops being classified on the host)
    want_surv = calling._step3_survivors(want2, 6)
    assert engine.table_bytes(engine.TABLE_STEP3_ROWS, n_surv) == want_surv, "survivors"
    return got2, want_surv


def device_tables(engine, names, ct_names, n_ct, tmp):
    engine.set_table_names(names, ct_names)
    out = {}
    for table in list(range(n_ct)) + [engine.TABLE_MERGED, engine.TABLE_STEP1, engine.TABLE_STEP1_KEPT]:
        n = engine.format_table(table)
        out[table] = engine.table_bytes(table, n)
        p = os.path.join(tmp, "dev%d.tsv" % table)
        with open(p, "wb") as f:
            f.write(b"#header\n")
        engine.append_table(table, p)                              # the streamed file: the header the caller wrote + the same bytes
        assert open(p, "rb").read() == b"#header\n" + out[table]
    engine.free_table()
    return out


def same_tables(got, want):
    assert sorted(got) == sorted(want)
    for t in sorted(want):
        if got[t] != want[t]:
            g, w = got[t].split(b"\n"), want[t].split(b"\n")
            bad = [(a, b) for a, b in zip(g, w) if a != b]
            raise AssertionError("table %d: %d / %d lines, first mismatch:\n%r\n%r" % (t, len(g), len(w), bad[0][0] if bad else None, bad[0][1] if bad else None))


@pytest.mark.parametrize("n_ct", [1, 2, 3, 4])
def test_tables_of_installed_counts(engine, tmp_path, n_ct):
    """count rows parsed from the reference's golden BaseCellCounter tables, different site sets per cell type (NA cells in the merged and
    step-1 rows, 1 to 4 cell types)"""
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    base = [tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.%s.tsv" % ct), names)[:3] for ct in ("Cancer", "Non-Cancer")]
    rng = np.random.default_rng(40 + n_ct)
    per_ct, ct_names = [], ["T%d" % i if i else "Cancer cells" for i in range(n_ct)]
    for i in range(n_ct):
        k, r, c = base[i % 2]
        keep = rng.random(len(k)) < 0.8
        per_ct.append((k[keep], r[keep], c[keep]))
    engine.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
    n_sites, n_cand = engine.call_step1(CallParams.longsom_defaults(min_cell_types=min(2, n_ct)))
    assert n_cand > 0
    calls = engine.fetch_calls()
    per_ct = [engine.fetch_counts(ct) for ct in range(n_ct)]                # (the REF column is the loaded reference's)
    want = host_tables(str(tmp_path), per_ct, calls, names, ct_names)
    same_tables(device_tables(engine, names, ct_names, n_ct, str(tmp_path)), want)
    # step 2 on the device: a third of the candidate sites in each position set (some in two)
    cand = np.ascontiguousarray(calls["key"][(calls["site_filter"] >> 31) != 0]) + 1
    possets = [np.sort(rng.choice(cand, len(cand) // 3, replace=False)) for _ in range(3)]
    s2, surv = step2_on_device_equals_host(engine, names, ct_names, want[6], possets)
    assert b"RNA_editing_db" in s2 and b"PoN_SR" in s2 and b",PoN_LR" in s2
    if n_ct > 1:
        assert b"\t\t" in s2 or s2.endswith(b"\t\n") or b"\t\n" in s2           # NA cells went blank


def test_contig_order_context_edges_and_rounding_ties(engine, tmp_path):
    """contigs whose Python string order is not their index order, sites in the first and last five bases of a contig (Up / Down context
    '.' or cut short), contigs without a row, and VAF / MCF quotients that sit on rounding ties: k/32 (an exact binary tie: half-even) and
    (2k+1)/20000 (a decimal tie the double quotient falls off to one side of)"""
    rng = np.random.default_rng(78)
    lens = [197, 70, 6, 64, 129, 11, 65, 63, 40]
    names = ["chr1", "chr10", "chr2", "chrM", "chr11", "chr3", "chrX", "chr20", "chr1_alt"]
    seqs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), L) for L in lens]
    engine.set_contigs(lens)
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    cls_of = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}
    per_ct = []
    for ct in range(2):
        keys, rows = [], []
        for t, L in enumerate(lens):
            if t == 5 or (t == 8 and ct == 0):
                continue                                            # a contig without rows; one only the second cell type has
            for pos in range(L):
                if rng.random() < 0.15:
                    continue
                r = np.zeros(42, np.uint32)
                rc = cls_of[int(seqs[t][pos])]
                bc = np.zeros(8, np.int64); cc = np.zeros(8, np.int64)
                kind = rng.integers(0, 4)
                if kind == 0:                                       # DP 32, NC 32: ties in binary
                    alts = rng.permutation([c for c in range(4) if c != rc])[:2]
                    bc[alts[0]], bc[alts[1]] = rng.choice([5, 7, 3]), rng.choice([1, 5])
                    bc[rc] = 32 - bc.sum(); cc[:] = np.minimum(bc, [5, 7, 3, 5, 0, 0, 0, 0]); nc = 32
                elif kind == 1:                                     # DP 20000, NC 2000
                    alt = rng.choice([c for c in range(4) if c != rc])
                    bc[alt] = 2 * rng.integers(500, 4000) + 1; bc[rc] = 20000 - bc[alt]
                    cc[alt] = 2 * rng.integers(20, 300) + 1; cc[rc] = 1500; nc = 2000
                else:
                    bc[rc] = rng.integers(15, 60)
                    for alt in rng.permutation([c for c in range(4) if c != rc])[:rng.integers(0, 3)]:
                        bc[alt] = rng.integers(1, 12)
                    cc = np.minimum(bc, rng.integers(1, 9, 8)) * (bc > 0); nc = max(5, int(cc.max()) + 3)
                r[0] = bc.sum(); r[1] = nc; r[2:10] = cc; r[10:18] = bc; r[18:26] = bc * 30; r[26:34] = bc // 2; r[34:42] = bc - bc // 2
                keys.append((t << 32) | pos); rows.append(r)
        per_ct.append((np.asarray(keys, np.int64), None, np.stack(rows)))
    ct_names = ["Cancer", "Non-Cancer"]
    engine.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
    n_sites, n_cand = engine.call_step1()
    calls = engine.fetch_calls()
    per_ct = [engine.fetch_counts(ct) for ct in range(2)]
    want = host_tables(str(tmp_path), per_ct, calls, names, ct_names)
    s1 = want[5].decode().split("\n")
    vaf = [x for l in s1 if l and l.split("\t")[4] != "." for x in l.split("\t")[14].replace("|", ",").split(",")]
    assert "0.1562" in vaf and "0.2188" in vaf and sum(1 for v in vaf if len(v) == 6) > 50, vaf[:20]
    assert any(l.split("\t")[7] == "." for l in s1 if l) and n_cand > 100
    order = [l.split("\t")[0] for l in want[4].decode().split("\n") if l]
    assert [c for i, c in enumerate(order) if i == 0 or order[i - 1] != c] == sorted(set(order)) and len(set(order)) == 8
    same_tables(device_tables(engine, names, ct_names, 2, str(tmp_path)), want)
    s2, surv = step2_on_device_equals_host(engine, names, ct_names, want[6], [np.zeros(0, np.int64)] * 3)
    assert len(surv) > 0 and any(l.startswith(b"chrM\t") for l in surv.split(b"\n"))


def test_tables_of_a_counted_sample(engine, tmp_path):
    """rows the count itself wrote (blocked planes, narrow and wide blocks) for a synthetic sample over the hg38 contig table"""
    from longsom_amd import synth
    m = synth.named("C2", n_reads=60_000, layout=1)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    engine.pileup_count()
    n_sites, n_cand = engine.call_step1()
    assert n_sites > 1000
    names = list(m.contig_names)
    ct_names = ["Cancer", "Non-Cancer"]
    calls = engine.fetch_calls()
    per_ct = [engine.fetch_counts(ct) for ct in range(2)]
    want = host_tables(str(tmp_path), per_ct, calls, names, ct_names)
    same_tables(device_tables(engine, names, ct_names, 2, str(tmp_path)), want)
    cand = np.ascontiguousarray(calls["key"][(calls["site_filter"] >> 31) != 0]) + 1
    step2_on_device_equals_host(engine, names, ct_names, want[6], [cand[::7], cand[::5], np.zeros(0, np.int64)])


def test_the_cell_classifier_on_odd_text(engine, tmp_path):
    """lsg_step2_summary's kinds of cell against tsvio.column_kinds on cells no table of this package holds: the classifier is the
    host's rule for rule, strtod's grammar included (cell-type names are printed as they are: they carry the odd text into column 6 and
    into the last columns' header-less cells)"""
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    k, r, c = tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.Cancer.tsv"), names)[:3]
    odd = ["1e5", "0x1F", "0x", "0x.8p-1", " 12", "12 ", "+5", "-0", "007", "1.50", "0.00001", "1234567890123456.5", ".5", "5.", "inf", "-inf", "Infinity", "infinit",
           "nan(abc_1)", "nan(", "NaN", "-nan", "N/A", "n/a", "None", "True", "false", "1,2", "1|2", "1e", "1e+", "0.0", "-0.0", "00.5", "1234567890123456789",
           "i", "I", "n", "N", ".", "-", "+", " ", "1.#IND", "<NA>", "0x1p", "1.5e3 ", "  7", "1_000", "٣"]
    odd += ["#N/A", "a#b", "#NA", "12#3"]
    engine.load_counts([k], [c])
    engine.call_step1()
    for kind in range(3):
        engine.load_posset(kind, np.zeros(0, np.int64))
    seen = set()
    for name in odd:
        engine.set_table_names(names, [name])
        n = engine.format_table(engine.TABLE_STEP2)
        text = engine.table_bytes(engine.TABLE_STEP2, n)
        assert ("\t%s\t" % name).encode() in text
        n_cols = 26
        kinds, _ = engine.step2_summary(n_cols)
        want_kinds = tsvio.column_kinds(text, n_cols)
        other = (want_kinds & tsvio.KIND_OTHER) != 0
        assert ((kinds & tsvio.KIND_OTHER) != 0).tolist() == other.tolist(), (name, kinds, want_kinds)
        assert kinds[~other].tolist() == want_kinds[~other].tolist(), (name, kinds, want_kinds)
        seen.add(int(want_kinds[6]))
    assert {tsvio.KIND_NA, tsvio.KIND_INT, tsvio.KIND_FLOAT, tsvio.KIND_ODD, tsvio.KIND_OTHER} <= seen, seen
    engine.free_table()


def test_table_calls_refuse_what_they_cannot_print(engine):
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    k, r, c = tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.Cancer.tsv"), names)[:3]
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    engine.load_counts([k], [c])
    with pytest.raises(RuntimeError, match="lsg_set_table_names"):
        engine.format_table(0)
    engine.set_table_names(names, ["Cancer"])
    with pytest.raises(RuntimeError, match="lsg_call_step1"):
        engine.format_table(engine.TABLE_MERGED)
    with pytest.raises(RuntimeError, match="no cell type"):
        engine.format_table(1)
    with pytest.raises(RuntimeError, match="not formatted"):
        engine.table_bytes(engine.TABLE_STEP1, 10)
    assert engine.format_table(0) > 0
    engine.free_table()


def test_the_chain_with_device_tables_writes_the_host_paths_files(engine, tmp_path, monkeypatch):
    """pipeline.run_snv end to end on a synthetic BAM (C2's model, 200 k reads, editing / PoN files that hit candidate sites): the run that
    prints its tables and runs step 2 on the device (the default) leaves, byte for byte, the eight files of the run that takes the host
    writers and the host's step 2 (LONGSOM_HOST_TABLES=1: csrc/hostio, pinned to the reference's goldens); so does the run that prints
    on the device but leaves step 2 to the host (LONGSOM_HOST_STEP2=1)"""
    from longsom_amd import hostio, pipeline, synth
    m = synth.named("C2", n_reads=200_000)
    bam, fa, bct = str(tmp_path / "S.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "bc.tsv")
    hostio.synth_bam(m, bam, fa)
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    engine.unload_reads()
    monkeypatch.setenv("LONGSOM_HOST_TABLES", "1")
    host = pipeline.run_snv(bam, bct, fa, str(tmp_path / "host0"), "S")
    # position-set files from the candidates the first run found (chrom, 1-based pos)
    rows = [l.split("\t")[:2] for l in open(host.step2).read().split("\n") if l and not l.startswith("#")]
    assert len(rows) > 1000
    files = {}
    for name, step in (("editing", 11), ("pon_sr", 7), ("pon_lr", 5)):
        files[name] = str(tmp_path / (name + ".tsv"))
        with open(files[name], "w") as f:
            f.write("#chrom\tpos\n" + "".join("%s\t%s\n" % (c, p) for c, p in rows[::step]))
    kw = dict(editing=files["editing"], pon_sr=files["pon_sr"], pon_lr=files["pon_lr"])
    host = pipeline.run_snv(bam, bct, fa, str(tmp_path / "host"), "S", **kw)
    monkeypatch.delenv("LONGSOM_HOST_TABLES")
    dev = pipeline.run_snv(bam, bct, fa, str(tmp_path / "dev"), "S", **kw)
    monkeypatch.setenv("LONGSOM_HOST_STEP2", "1")
    mixed = pipeline.run_snv(bam, bct, fa, str(tmp_path / "mixed"), "S", **kw)
    assert "format_tables" in dev.timings and "format_tables" in mixed.timings and "format_tables" not in host.timings
    def files_of(o):
        return list(o.counts.values()) + [o.merged, o.step1, o.step2, o.step3, o.step3_unfiltered]
    for a, b, c in zip(files_of(host), files_of(dev), files_of(mixed)):
        want = open(a, "rb").read()
        assert len(want) > 200 and os.path.basename(a) == os.path.basename(b)
        assert open(b, "rb").read() == want, os.path.basename(a)
        assert open(c, "rb").read() == want, os.path.basename(a) + " (host step 2)"
    s2 = open(host.step2).read()
    assert "RNA_editing_db" in s2 and "PoN_SR" in s2 and "PoN_LR" in s2
    assert sum(1 for l in open(host.step3_unfiltered) if not l.startswith("#")) > 10
