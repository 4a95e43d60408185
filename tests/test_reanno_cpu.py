"""Host side of the re-annotation pass against golden files produced by RUNNING the reference's
own code (tools/make_goldens.py --reanno-only): HighConfidenceCancerVariants.py and CellTypeReannotation.py; plus the
events-level genotype oracle against hand-derived known answers."""
import filecmp
import os
import shutil

import numpy as np
import pytest

from longsom_amd import reanno
from longsom_amd.engine import ReadRecords

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag,args", [("sample", (50, 0.2, 0.25, 10000)), ("sample.loose", (5, 0.05, 0.1, 400))])
def test_hccv_filter_equals_reference(tmp_path, tag, args):
    out = reanno.hccv_filter(os.path.join(G, "sample.calling.step2.tsv"), str(tmp_path / tag), *args)
    for suffix in ("", "2", "3"):
        got, want = open(out + suffix).read(), open(os.path.join(G, tag + ".HCCV.tsv" + suffix)).read()
        assert got == want, "%s.HCCV.tsv%s differs" % (tag, suffix)
    n = sum(1 for l in open(out) if not l.startswith("#"))
    assert n >= 3                                        # the fixture keeps real rows through every filter


@pytest.mark.parametrize("seed", range(4))
def test_hccv_filter_column_form_equals_the_row_form(tmp_path, seed, monkeypatch):
    """the column-wise hccv_filter (row functions only where they can change a row) against the reference-shaped row-wise one — which the
    goldens above pin — on tables made of the golden's rows: replicated with shifted positions, some moved to chrM, and cut down to one
    kind of row (single cell type, Cancer only, multi-allelic only, none multi-allelic: the column dtypes pandas infers differ)"""
    import filecmp
    lines = open(os.path.join(G, "sample.calling.step2.tsv")).read().split("\n")
    head = [l for l in lines if l.startswith("#")]
    rows = [l.split("\t") for l in lines if l and not l.startswith("#")]
    rng = np.random.default_rng(seed)
    cases = [("all", None), ("single", lambda r: "," not in r[6]), ("cancer_only", lambda r: r[6] == "Cancer"), ("multi", lambda r: "Multi" in r[5]),
             ("nomulti", lambda r: "Multi" not in r[5] and "|" not in r[4]), ("noncancer", lambda r: r[6] == "Non-Cancer")]
    args = [(50, 0.2, 0.25, 10000), (5, 0.05, 0.1, 400), (20, 0.1, 0.4, 10000)][seed % 3]
    for name, pick in cases:
        out = list(head)
        for k in range(int(rng.integers(1, 4))):
            for r in (rows if pick is None else [r for r in rows if pick(r)]):
                if rng.random() < 0.15:
                    continue
                r2 = list(r)
                shift = k * 1_000_000 + int(rng.integers(0, 3)) * 7
                r2[1] = str(int(r[1]) + shift); r2[2] = str(int(r[2]) + shift)
                if rng.random() < 0.05:
                    r2[0] = "chrM"
                out.append("\t".join(r2))
        table = tmp_path / ("%s.tsv" % name)
        table.write_text("\n".join(out) + "\n")
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("LONGSOM_HCCV_ROW_PATH", mode)
            try:
                got[mode] = reanno.hccv_filter(str(table), str(tmp_path / ("%s.%s" % (name, mode))), *args)
            except Exception as e:                            # noqa: BLE001 - (a table with no row left: the reference's frame operations raise; both forms must)
                got[mode] = type(e)
        if isinstance(got["1"], type) or isinstance(got["0"], type):
            assert got["1"] == got["0"], (name, got)
            continue
        for sfx in ("", "2", "3"):
            assert filecmp.cmp(got["1"] + sfx, got["0"] + sfx, shallow=False), (name, sfx)
        if name == "all":
            assert sum(1 for l in open(got["0"]) if not l.startswith("#")) > 3


@pytest.mark.parametrize("tag,fusions,mv,mf", [("reanno", "reanno.Fusions.SingleCellGenotype.tsv", 3, 0.25), ("reanno.nofusion", "", 2, 0.5)])
def test_celltype_reannotation_equals_reference(tmp_path, tag, fusions, mv, mf):
    out = str(tmp_path / "out.tsv")
    kept, cancer = reanno.celltype_reannotation(os.path.join(G, "reanno.SNVs.SingleCellGenotype.tsv"), os.path.join(G, fusions) if fusions else "",
                                                os.path.join(G, "reanno.barcodes.tsv"), out, mv, mf)
    assert open(out).read() == open(os.path.join(G, tag + ".ReannotatedCellTypes.tsv")).read()
    assert 0 < cancer < kept <= 60


def test_cli_shims_write_the_same_files(tmp_path):
    from longsom_amd import cli
    cli.hccv(["--SNVs", os.path.join(G, "sample.calling.step2.tsv"), "--outfile", str(tmp_path / "s"), "--min_dp", "50", "--deltaVAF", "0.2",
              "--deltaMCF", "0.25", "--clust_dist", "10000"])
    assert filecmp.cmp(str(tmp_path / "s.HCCV.tsv"), os.path.join(G, "sample.HCCV.tsv"), shallow=False)
    cli.celltype_reannotation(["--SNVs", os.path.join(G, "reanno.SNVs.SingleCellGenotype.tsv"), "--fusions", os.path.join(G, "reanno.Fusions.SingleCellGenotype.tsv"),
                               "--outfile", str(tmp_path / "r.tsv"), "--meta", os.path.join(G, "reanno.barcodes.tsv"), "--min_variants", "3", "--min_frac", "0.25"])
    assert filecmp.cmp(str(tmp_path / "r.tsv"), os.path.join(G, "reanno.ReannotatedCellTypes.tsv"), shallow=False)


def test_target_windows_follow_the_reference_grouping(tmp_path):
    p = tmp_path / "v.tsv"
    p.write_text("##x\n#CHROM\tStart\nChr\tfoo\nchr1\t49999\tx\nchr1\t50000\tx\nchr1\t10\tx\nchr2\t50001\tx\n")
    g = reanno.read_target_windows(str(p), 50000)
    assert list(g) == ["chr1_0", "chr1_1", "chr2_1"]
    assert [el[1] for el in g["chr1_0"]] == ["49999", "10"]


# ---- events-level genotype oracle: hand-derived known answers ------------------------------------------------------
def _ev(sym, q):
    return 0 if sym >= 8 else (0x0800 | (sym << 8) | q)


def test_genotype_oracle_known_answers():
    from oracle import genotype_oracle as go
    # one contig of 100 bp; 3 barcodes (cb 2 is not in barcodes.tsv -> celltype 255); target sites at pos0 10 and 12
    celltype_of = np.array([0, 1, 255], np.uint8)
    # reads: r0 cb0 fwd MQ60 covering 8..13 : syms  A  C  G(q10) T  'O'(q40) N
    #        r1 cb0 rev MQ60 covering 10..11: syms  G  G
    #        r2 cb1 MQ20 (fails min_mq 60) covering 10: G
    #        r3 cb1 supplementary covering 10: G
    #        r4 cb2 (unknown cell) covering 10: G
    #        r5 cb1 ok covering 12 with an insertion anchor I(q35), and 10 with 'NA'
    #        r6 cb1 ok, raw CB carried "-1" (flag bit 15) covering 10: G
    seg = [(0, 8, [_ev(0, 40), _ev(1, 40), _ev(3, 10), _ev(2, 40), _ev(7, 40), _ev(6, 40)]),
           (1, 10, [_ev(3, 40), _ev(3, 40)]),
           (2, 10, [_ev(3, 40)]),
           (3, 10, [_ev(3, 40)]),
           (4, 10, [_ev(3, 40)]),
           (5, 10, [_ev(15, 40), _ev(0, 40), _ev(4, 35)]),
           (6, 10, [_ev(3, 40)])]
    flags = np.array([0, 16, 0, 0x800, 0, 0, 0x8000], np.uint16)
    mapq = np.array([60, 60, 20, 60, 60, 60, 60], np.uint8)
    cb = np.array([0, 0, 1, 1, 2, 1, 1], np.int32)
    events, off, start, ln, rd = [], [], [], [], []
    for r, st, evs in seg:
        rd.append(r); start.append(st); ln.append(len(evs)); off.append(len(events)); events += evs
    rec = ReadRecords(np.zeros(7, np.int32), np.array([s[1] for s in seg], np.int32), flags, mapq, cb, np.array(rd, np.uint32),
                      np.array(start, np.int32), np.array(ln, np.int32), np.array(off, np.int64), np.array(events, np.uint16))
    keys = np.array([10, 12], np.int64)
    alt = np.array([3, 4], np.uint8)                       # expected alts: G at 10, I at 12
    dp, al = go.genotype(rec, np.array([100]), celltype_of, keys, alt, min_bq=30, min_mq=60)
    # site 10: r0's G has quality 10 (< 30) -> dropped; r1 counts (G = alt); r2/r3/r4 fail admission; r5 'NA'; r6 strict CB
    assert dp[0].tolist() == [1, 0, 0] and al[0].tolist() == [1, 0, 0]
    # site 12: r0's 'O' is not a base; r5's I counts as alt for cb1
    assert dp[1].tolist() == [0, 1, 0] and al[1].tolist() == [0, 1, 0]
    # without the strict CB rule r6 counts; with min_bq 0 r0's G does too
    dp, al = go.genotype(rec, np.array([100]), celltype_of, keys, alt, min_bq=0, min_mq=60, strict_cb=0)
    assert dp[0].tolist() == [2, 1, 0] and al[0].tolist() == [2, 1, 0]
    # --alt_flag Alt at a site whose expected alt is T: only reads carrying T are looked at
    dp, al = go.genotype(rec, np.array([100]), celltype_of, np.array([11], np.int64), np.array([2], np.uint8), min_bq=30, min_mq=60, alt_only=1)
    assert dp[0].tolist() == [1, 0, 0] and al[0].tolist() == [1, 0, 0]


def test_cell_row_text():
    from oracle import genotype_oracle as go
    a2, b2 = 0.260288007167716, 173.94711910763732
    assert go.cell_row("chr1", 9, "A", "G", "Cancer", "7", "ACGT", "Cancer", 0, 0, a2, b2, 0.01, "True").endswith("\t0\t0\t.\t.\tNoCoverage")
    assert go.cell_row("chr1", 9, "A", "G", "Cancer", "7", "ACGT", "Cancer", 4, 0, a2, b2, 0.01, "True").endswith("\t4\t0\t0.0\t.\tNoAltReads")
    assert go.cell_row("chrM", 9, "A", "G", "Cancer", "7", "ACGT", "Cancer", 10, 2, a2, b2, 0.01, "True").endswith("\t10\t2\t0.2\t.\tLowVAFChrM")
    assert go.cell_row("chrM", 9, "A", "G", "Cancer", "7", "ACGT", "Cancer", 10, 2, a2, b2, 0.01, "False").split("\t")[-1] in ("PASS", "BetaBin_problem")
    r = go.cell_row("chr1", 9, "A", "G", "Cancer", "7", "ACGT", "Cancer", 10, 5, a2, b2, 0.01, "True").split("\t")
    assert r[:3] == ["chr1", "10", "10"] and r[11] == "0.5" and r[13] == "PASS"


def test_sf_table_is_scipy():
    """the golden per-cell tail table is what scipy gives here (guards the fixture, not the product)"""
    import json
    from scipy.stats import betabinom
    t = json.load(open(os.path.join(G, "betabinom_sf_table.json")))
    for n, k, txt in t["rows"]:
        assert str(round(betabinom.sf(k - 0.001, n, t["alpha2"], t["beta2"]), 4)) == txt
