"""The pileup half against files WRITTEN BY THE REFERENCE'S OWN CODE (tools/make_pileup_goldens.py ran
SplitBamCellTypes.py and BaseCellCounter.py from /root/reference over column-replay stand-ins for pysam / pybedtools
and committed their outputs under tests/golden/pileup.*).  What these fixtures pin: meta_to_dict and split_bam's routing
and report (a1, a2), MakeWindows / position 0 / the 50 001 window edge (a3), EasyReadPileup's symbol mapping (a6), the
counting loop (a7), the gates and the row text (a8), temp files and the file order chr1, chr10, chr2, chrM (a9).  Only the
CIGAR -> column step (a4-a5: htslib / pysam, absent from the reference tree) stays hand-derived; it is stated three times
independently (tools/minipysam.py, oracle/plp_oracle.c, csrc/hostio/bamio.cpp) and all three must agree here.

CPU (not gpu): both C oracles and the decoder reproduce the reference's bytes.  GPU: the HIP path does."""
import json
import os

import numpy as np
import pytest

from longsom_amd import hostio, tsvio
from tests.support import bamwrite
from longsom_amd._lib import CountParams
from oracle import loader
from tests import kat_pileup_cases as K

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(G, "pileup.kat.json")))
DATE = "##fileDate=x\n"


def no_date(text):
    return "".join(l for l in text.splitlines(True) if not l.startswith("##fileDate="))


def table(keys, refs, counts, names, sample_id):
    """the product's text form of one cell type's rows, minus the date line; None when there is no row (the reference then
    writes no file at all: 'No temporary files found', BaseCellCounter.py:76-79)"""
    if len(keys) == 0:
        return None
    return no_date(tsvio.format_counts_tsv(keys, refs, counts, names, sample_id, DATE))


def report_text(rep):
    keys = ["Total_reads", "Pass_reads", "CB_not_found", "CB_not_matched"] + (["MAPQ"] if "MAPQ" in rep else [])
    return "\t".join(keys) + "\n" + "\t".join(str(rep[k]) for k in keys) + "\n"


# ---- known-answer cases: reference-written tables == hand-derived expectations == oracles ------------------------
KREF = np.frombuffer(K.REF.encode(), dtype=np.uint8)
KBC = [b for b, _ in K.BARCODES]
KCT = np.array([0 if t == "Cancer" else 1 for _, t in K.BARCODES], np.uint8)


def kat_bam(tmp_path, name):
    p = str(tmp_path / (name + ".bam"))
    bamwrite.write_bam(p, [K.CONTIG], sorted(K.CASES[name]["reads"], key=lambda r: r["pos"]))
    return p


@pytest.mark.parametrize("name", sorted(K.CASES))
def test_reference_tables_equal_hand_derived_rows(name):
    """the reference's own code, fed the columns of the known-answer records, prints what was worked out by hand"""
    for ct, key in (("Cancer", "cancer"), ("Non-Cancer", "noncancer")):
        want = K.expected_rows(K.CASES[name].get(key, {}))
        text = KAT[name]["tables"].get(ct)
        got = {}
        for line in (text or "").split("\n"):
            if line and not line.startswith("#"):
                f = line.split("\t")
                dp, nc, cc, bc, bq, bcf, bcr = f[4].split("|")
                row = [int(dp), int(nc)] + [int(x) for v in (cc, bc, bq, bcf, bcr) for x in v.split(":")]
                got[int(f[1]) - 1] = row
        assert sorted(got) == sorted(want)
        for pos, row in want.items():
            assert got[pos] == [row[i] for i in K.PRINTED], (name, ct, pos + 1)


@pytest.mark.parametrize("name", sorted(K.CASES))
def test_oracles_write_the_reference_tables_kat(tmp_path, name):
    bam = kat_bam(tmp_path, name)
    p = KAT[name]["params"]
    dec = hostio.decode_bam(bam, KBC, min_mapq=p["min_mq"])
    assert report_text(dec.report) == KAT[name]["report"]
    for ct, cname in enumerate(("Cancer", "Non-Cancer")):
        want = KAT[name]["tables"].get(cname)
        k, r, c = loader.plp_count(bam, KBC, KCT, ct, [K.CONTIG[1]], [KREF], **p)
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == want, "plp_oracle"
        k, r, c, _ = loader.count(dec.records, [K.CONTIG[1]], [KREF], KCT, ct, p["min_bq"], p["min_mq"], p["min_dp"], p["min_cc"])
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == want, "decoder + count_oracle"


# ---- htslib <= 1.10: the one place where the CIGAR -> column step depends on the library version --------------------------------
KAT_LEGACY = json.load(open(os.path.join(G, "pileup.kat_legacy.json")))


@pytest.fixture
def legacy_del_merge():
    """htslib <= 1.10 mode of the decoder and of the BAM-level oracle for the duration of a test"""
    old = hostio.set_legacy_del_merge(True)
    loader.plp_set_legacy_del_merge(True)
    yield
    hostio.set_legacy_del_merge(old)
    loader.plp_set_legacy_del_merge(False)


def parse_table(text):
    got = {}
    for line in (text or "").split("\n"):
        if line and not line.startswith("#"):
            f = line.split("\t")
            dp, nc, cc, bc, bq, bcf, bcr = f[4].split("|")
            got[int(f[1]) - 1] = [int(dp), int(nc)] + [int(x) for v in (cc, bc, bq, bcf, bcr) for x in v.split(":")]
    return got


@pytest.mark.parametrize("name", sorted(K.LEGACY_CASES))
def test_legacy_del_merge_tables(tmp_path, name, legacy_del_merge):
    """inside "1D2D" htslib <= 1.10 flags the first deletion's column as a deletion anchor: the reference (over the stand-in's legacy
    mode) prints a D there, htslib >= 1.11 an O; the hand-derived rows, the reference-written table, both oracles and the decoder agree
    in BOTH modes (the default mode is the ordinary consecutive_deletions case)"""
    want_rows = K.expected_rows(K.LEGACY_CASES[name]["cancer"])
    got = parse_table(KAT_LEGACY[name]["tables"].get("Cancer"))
    assert sorted(got) == sorted(want_rows)
    for pos, row in want_rows.items():
        assert got[pos] == [row[i] for i in K.PRINTED], (name, pos + 1)
    assert KAT_LEGACY[name]["tables"]["Cancer"] != KAT[name]["tables"]["Cancer"]              # the version dependence is real
    bam = str(tmp_path / (name + ".bam"))
    bamwrite.write_bam(bam, [K.CONTIG], sorted(K.LEGACY_CASES[name]["reads"], key=lambda r: r["pos"]))
    p = KAT_LEGACY[name]["params"]
    dec = hostio.decode_bam(bam, KBC, min_mapq=p["min_mq"])
    for ct, cname in enumerate(("Cancer", "Non-Cancer")):
        want = KAT_LEGACY[name]["tables"].get(cname)
        k, r, c = loader.plp_count(bam, KBC, KCT, ct, [K.CONTIG[1]], [KREF], **p)
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == want, "plp_oracle (legacy)"
        k, r, c, _ = loader.count(dec.records, [K.CONTIG[1]], [KREF], KCT, ct, p["min_bq"], p["min_mq"], p["min_dp"], p["min_cc"])
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == want, "decoder + count_oracle (legacy)"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(K.LEGACY_CASES))
def test_gpu_writes_the_legacy_tables(tmp_path, engine, name, legacy_del_merge):
    bam = str(tmp_path / (name + ".bam"))
    bamwrite.write_bam(bam, [K.CONTIG], sorted(K.LEGACY_CASES[name]["reads"], key=lambda r: r["pos"]))
    p = KAT_LEGACY[name]["params"]
    dec = hostio.decode_bam(bam, KBC, min_mapq=p["min_mq"])
    engine.set_contigs([K.CONTIG[1]]); engine.load_reference(0, KREF); engine.set_barcodes(KCT, 2); engine.set_region()
    engine.load_reads(dec.records)
    engine.pileup_count(CountParams.longsom_defaults(**p))
    for ct, cname in enumerate(("Cancer", "Non-Cancer")):
        k, r, c = engine.fetch_counts(ct)
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == KAT_LEGACY[name]["tables"].get(cname)


# ---- random multi-contig sample --------------------------------------------------------------------------------
def rand_inputs(tag):
    bc = hostio.read_barcodes(os.path.join(G, "pileup.%s.barcodes.tsv" % tag))
    names, seqs = tsvio.read_fasta(os.path.join(G, "pileup.rand.fa"))
    refs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), dtype=np.uint8) for s in seqs]
    return bc, names, refs


@pytest.mark.parametrize("tag", ["rand", "randsfx"])
def test_oracles_write_the_reference_tables_rand(tag):
    bc, names, refs = rand_inputs(tag)
    assert bc.celltype_names == ["Cancer", "Non-Cancer"]
    bam = os.path.join(G, "pileup.%s.bam" % tag)
    dec = hostio.decode_bam(bam, bc.barcodes, min_mapq=60)
    assert dec.contig_names == names == ["chr1", "chr10", "chr2", "chrM"]
    assert report_text(dec.report) == open(os.path.join(G, "pileup.%s.report.txt" % tag)).read()
    lens = [len(r) for r in refs]
    for ct, cname in enumerate(bc.celltype_names):
        want = open(os.path.join(G, "pileup.%s.%s.tsv" % (tag, cname))).read()
        k, r, c = loader.plp_count(bam, bc.barcodes, bc.celltype_of, ct, lens, refs)
        assert table(k, r, c, names, "s." + cname) == want, "plp_oracle"
        k, r, c, _ = loader.count(dec.records, lens, refs, bc.celltype_of, ct)
        assert table(k, r, c, names, "s." + cname) == want, "decoder + count_oracle"
    # the fixture does what it was built for
    text = open(os.path.join(G, "pileup.%s.Cancer.tsv" % tag)).read()
    chroms = [l.split("\t")[0] for l in text.split("\n") if l and not l.startswith("#")]
    assert sorted(set(chroms), key=chroms.index) == ["chr1", "chr10", "chr2", "chrM"]             # python string order of the windows
    pos1 = {int(l.split("\t")[1]) for l in text.split("\n") if l.startswith("chr1\t")}
    assert {50000, 50001, 50002} <= pos1 and 1 not in pos1 and 2 in pos1                              # window edge, position 0 skipped


# ---- depth cap: bam.pileup(..., max_depth) as the reference's run saw it (the stand-in applied htslib's bam_plp_push rule) ----------
# pileup.capw: a capped pile that STRADDLES the 50 001 window edge - the reference opens a pileup per 50 kb window (BaseCellCounter.py:185-191), so
# reads the first window's buffer drops are counted from 50 001 on by the second one's, which never held the short reads before the edge
CAP_CASES = [(8, "pileup.cap.bam", "pileup.cap.Cancer.tsv"), (200000, "pileup.cap.bam", "pileup.capoff.Cancer.tsv"),
             (8, "pileup.capw.bam", "pileup.capw.Cancer.tsv"), (200000, "pileup.capw.bam", "pileup.capwoff.Cancer.tsv")]


@pytest.mark.parametrize("max_depth,bam,golden", CAP_CASES)
def test_bam_level_oracle_applies_the_depth_cap(max_depth, bam, golden):
    bc, names, refs = rand_inputs("rand")
    k, r, c = loader.plp_count(os.path.join(G, bam), bc.barcodes, bc.celltype_of, 0, [len(x) for x in refs], refs, max_depth=max_depth)
    assert table(k, r, c, names, "s.Cancer") == open(os.path.join(G, golden)).read()
    assert open(os.path.join(G, "pileup.cap.Cancer.tsv")).read() != open(os.path.join(G, "pileup.capoff.Cancer.tsv")).read()
    if bam == "pileup.capw.bam" and max_depth == 8:
        # ONE pileup per contig (the cap replayed as a single stream) counts these columns differently: the fixture tells the two apart
        k, r, c = loader.plp_count(os.path.join(G, bam), bc.barcodes, bc.celltype_of, 0, [len(x) for x in refs], refs, max_depth=max_depth, window=0)
        assert table(k, r, c, names, "s.Cancer") != open(os.path.join(G, golden)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("max_depth,bam,golden", CAP_CASES)
def test_gpu_applies_the_depth_cap(engine, max_depth, bam, golden):
    bc, names, refs = rand_inputs("rand")
    dec = hostio.decode_bam(os.path.join(G, bam), bc.barcodes, min_mapq=60)
    engine.set_contigs([len(r) for r in refs])
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(bc.celltype_of, 2); engine.set_region()
    engine.load_reads(dec.records)
    engine.pileup_count(CountParams.longsom_defaults(max_depth=max_depth))
    k, r, c = engine.fetch_counts(0)
    assert table(k, r, c, names, "s.Cancer") == open(os.path.join(G, golden)).read()


# ---- per-cell genotyping: the reference's HCCVSingleCellGenotype.py over the same stand-in -------------------
def genotype_targets():
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "pileup.rand.HCCV.tsv")) if not l.startswith("#")]
    return rows


@pytest.mark.parametrize("tag,flag", [("rand", "All"), ("rand", "Alt"), ("randsfx", "All")])
def test_genotype_oracle_writes_the_reference_rows(tag, flag):
    from oracle import genotype_oracle as go
    from longsom_amd.reanno import SYM_OF_BASE
    bc, names, refs = rand_inputs(tag)
    dec = hostio.decode_bam(os.path.join(G, "pileup.%s.bam" % tag), bc.barcodes, min_mapq=0)
    tid_of = {n: i for i, n in enumerate(names)}
    sites = {}
    for r in genotype_targets():                                  # a later line naming the same site wins (:105)
        sites[(tid_of[r[0]] << 32) | (int(r[1]) - 1)] = r
    keys = np.asarray(sorted(sites), np.int64)
    alt_sym = np.asarray([SYM_OF_BASE.get(sites[k][4].split(",")[0], 255) for k in keys.tolist()], np.uint8)
    dp, alt = go.genotype(dec.records, [len(r) for r in refs], bc.celltype_of, keys, alt_sym, min_bq=30, min_mq=60, alt_only=1 if flag == "Alt" else 0)
    got = set()
    for i, k in enumerate(keys.tolist()):
        r = sites[k]
        for b, name in enumerate(bc.barcodes):
            got.add(go.cell_row(r[0], k & 0xFFFFFFFF, r[3], r[4].split(",")[0], r[6], r[13], name, bc.celltype_names[int(bc.celltype_of[b])], int(dp[i, b]), int(alt[i, b]),
                                0.260288007167716, 173.94711910763732, 0.01, "True"))
    want = [l for l in open(os.path.join(G, "pileup.%s.genotype.%s.tsv" % (tag, flag))).read().split("\n")[1:] if l]
    assert len(want) == len(got) == len(keys) * len(bc.barcodes) and set(want) == got
    assert (tag == "randsfx") == (not any(l.split("\t")[9] != "0" for l in want))      # raw "-1" CBs are not keys of the cleaned table (:160-161)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,flag", [("rand", "All"), ("rand", "Alt"), ("randsfx", "All")])
def test_gpu_genotype_table_equals_the_reference_file(engine, tmp_path, tag, flag):
    from longsom_amd import reanno
    bc, names, refs = rand_inputs(tag)
    dec = hostio.decode_bam(os.path.join(G, "pileup.%s.bam" % tag), bc.barcodes, min_mapq=0)
    engine.set_contigs([len(r) for r in refs])
    engine.set_barcodes(bc.celltype_of, 2); engine.set_region()
    engine.load_reads(dec.records)
    out = str(tmp_path / "geno.tsv")
    n = reanno.single_cell_genotype(engine, os.path.join(G, "pileup.rand.HCCV.tsv"), bc, names, out, alt_flag=flag, min_bq=30, min_mq=60)
    want = open(os.path.join(G, "pileup.%s.genotype.%s.tsv" % (tag, flag))).read()
    assert n == want.count("\n") - 1
    assert open(out).read() == want


# ---- the genotyping pileup's buffer holds reads WITHOUT a listed barcode too (HCCVSingleCellGenotype.py:121-122 piles up the unsplit BAM):
# tests/golden/pileup.capu.* (tools/make_genotype_cap_golden.py: the reference run with max_depth = 8 on a pile of mostly unlisted reads)
def capu_sites(names):
    tid_of = {n: i for i, n in enumerate(names)}
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "pileup.capu.HCCV.tsv")) if not l.startswith("#")]
    sites = {(tid_of[r[0]] << 32) | (int(r[1]) - 1): r for r in rows}
    return np.asarray(sorted(sites), np.int64), sites


@pytest.mark.parametrize("keep", [True, False])
def test_unlisted_reads_fill_the_genotyping_buffer_oracle(keep):
    from oracle import genotype_oracle as go
    from longsom_amd.reanno import SYM_OF_BASE
    bc, names, refs = rand_inputs("rand")
    old = hostio.set_keep_unlisted(keep)
    try:
        dec = hostio.decode_bam(os.path.join(G, "pileup.capu.bam"), bc.barcodes, min_mapq=0)
    finally:
        hostio.set_keep_unlisted(old)
    assert dec.records.n_reads == (57 if keep else 27)
    keys, sites = capu_sites(names)
    alt_sym = np.asarray([SYM_OF_BASE.get(sites[k][4].split(",")[0], 255) for k in keys.tolist()], np.uint8)
    got = {}
    for tag, cap in (("capu", 8), ("capuoff", 200000)):
        dp, alt = go.genotype(dec.records, [len(r) for r in refs], bc.celltype_of, keys, alt_sym, min_bq=30, min_mq=60,
                              group_off=np.asarray([0, len(keys)], np.int64), max_depth=cap)
        rows = set()
        for i, k in enumerate(keys.tolist()):
            r = sites[k]
            for b, name in enumerate(bc.barcodes):
                rows.add(go.cell_row(r[0], k & 0xFFFFFFFF, r[3], r[4].split(",")[0], r[6], r[13], name, bc.celltype_names[int(bc.celltype_of[b])], int(dp[i, b]), int(alt[i, b]),
                                     0.260288007167716, 173.94711910763732, 0.01, "True"))
        want = {l for l in open(os.path.join(G, "pileup.%s.genotype.All.tsv" % tag)).read().split("\n")[1:] if l}
        got[tag] = rows == want
    assert got["capuoff"]                                          # nothing dropped: the unlisted reads do not matter
    assert got["capu"] == keep                                     # capped: only a pool that holds them drops what the reference drops


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["host", "device"])
def test_gpu_genotype_buffer_holds_unlisted_reads(engine, tmp_path, ingest):
    from longsom_amd import reanno
    bc, names, refs = rand_inputs("rand")
    bam = os.path.join(G, "pileup.capu.bam")
    engine.set_contigs([len(r) for r in refs])
    engine.set_barcodes(bc.celltype_of, 2); engine.set_region()
    texts = {}
    for keep in (True, False):
        if ingest == "host":
            old = hostio.set_keep_unlisted(keep)
            try:
                dec = hostio.decode_bam(bam, bc.barcodes, min_mapq=0)
            finally:
                hostio.set_keep_unlisted(old)
            engine.load_reads(dec.records)
        else:
            for t, r in enumerate(refs):
                engine.load_reference(t, r)
            engine.set_keep_unlisted(keep)
            try:
                info, _, _ = engine.load_bam(bam, bc.barcodes, min_mapq=0, first_record_offset=hostio.bam_header(bam)[2])
            finally:
                engine.set_keep_unlisted(False)
            assert int(info["total_reads"]) == 57 and int(info["cb_not_found"]) + int(info["cb_not_matched"]) == 30
        assert engine.store_shape()[0] > 0
        for tag, cap in (("capu", 8), ("capuoff", 200000)):
            out = str(tmp_path / ("%s_%d.tsv" % (tag, keep)))
            reanno.single_cell_genotype(engine, os.path.join(G, "pileup.capu.HCCV.tsv"), bc, names, out, alt_flag="All", min_bq=30, min_mq=60, max_depth=cap)
            texts[(tag, keep)] = open(out).read()
    for tag in ("capu", "capuoff"):
        want = open(os.path.join(G, "pileup.%s.genotype.All.tsv" % tag)).read()
        assert texts[(tag, True)] == want, tag
    assert texts[("capuoff", False)] == open(os.path.join(G, "pileup.capuoff.genotype.All.tsv")).read()
    assert texts[("capu", False)] != texts[("capu", True)]           # without the unlisted reads the buffer never fills


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(K.CASES))
def test_gpu_writes_the_reference_tables_kat(tmp_path, engine, name):
    p = KAT[name]["params"]
    dec = hostio.decode_bam(kat_bam(tmp_path, name), KBC, min_mapq=p["min_mq"])
    engine.set_contigs([K.CONTIG[1]]); engine.load_reference(0, KREF); engine.set_barcodes(KCT, 2); engine.set_region()
    engine.load_reads(dec.records)
    engine.pileup_count(CountParams.longsom_defaults(**p))
    for ct, cname in enumerate(("Cancer", "Non-Cancer")):
        k, r, c = engine.fetch_counts(ct)
        assert table(k, r, c, [K.CONTIG[0]], "kat." + cname) == KAT[name]["tables"].get(cname)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["rand", "randsfx"])
def test_gpu_writes_the_reference_tables_rand(engine, tag):
    bc, names, refs = rand_inputs(tag)
    dec = hostio.decode_bam(os.path.join(G, "pileup.%s.bam" % tag), bc.barcodes, min_mapq=60)
    engine.set_contigs([len(r) for r in refs])
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(bc.celltype_of, 2); engine.set_region()
    engine.load_reads(dec.records)
    engine.pileup_count(CountParams.longsom_defaults())
    for ct, cname in enumerate(bc.celltype_names):
        k, r, c = engine.fetch_counts(ct)
        assert table(k, r, c, names, "s." + cname) == open(os.path.join(G, "pileup.%s.%s.tsv" % (tag, cname))).read()


@pytest.mark.gpu
def test_gpu_pipeline_files_equal_reference_chain(tmp_path):
    """the fused rule (BAM + barcodes.tsv + FASTA in) writes the SplitBam report and the per-cell-type tables the reference chain wrote"""
    from longsom_amd import pipeline
    out = pipeline.run_snv(os.path.join(G, "pileup.rand.bam"), os.path.join(G, "pileup.rand.barcodes.tsv"), os.path.join(G, "pileup.rand.fa"),
                           str(tmp_path), "s")
    for cname in ("Cancer", "Non-Cancer"):
        assert no_date(open(out.counts[cname]).read()) == open(os.path.join(G, "pileup.rand.%s.tsv" % cname)).read()
    rep = open(out.report).read().split("\n")
    want = open(os.path.join(G, "pileup.rand.report.txt")).read().split("\n")
    h, r = rep[0].split("\t"), rep[1].split("\t")
    assert h[-1] == "Total_time" and "\t".join(h[:-1]) == want[0] and "\t".join(r[:-1]) == want[1]
