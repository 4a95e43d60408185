"""GPU end-to-end: BAM + barcodes.tsv + FASTA -> every TSV of the SNV chain, byte-compared with the oracle chain
(BAM-level pileup oracle -> merge -> step1 -> step2 restatements; step3 is pinned separately by the reference goldens)."""
import os

import numpy as np
import pytest

from tests.util import neg_zero

from longsom_amd import hostio, pipeline, synth, tsvio
from oracle import calling_oracle as co
from oracle import loader

pytestmark = pytest.mark.gpu


def strip_date(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("##fileDate="))


def test_fused_chain_matches_oracle_chain(tmp_path, engine):
    m = synth.named("C1", n_reads=6000, n_genes=30, n_cb=80, snp_mod=300)
    bam, fa, bct = str(tmp_path / "S1.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa, barcode_suffix="-1")
    bcs = hostio.synth_barcodes(m)
    hostio.write_barcodes_tsv(bct, bcs, m.celltype_of, ["Cancer", "Non-Cancer"], suffix="-1")
    names, refs = tsvio.read_fasta(fa)
    # oracle chain
    texts, rows = [], []
    for ct, ctn in enumerate(("Cancer", "Non-Cancer")):
        k, r, c = loader.plp_count(bam, bcs, m.celltype_of, ct, m.contig_len, refs)
        rows.append((k, r, c))
        texts.append(tsvio.format_counts_tsv(k, r, c, names, "S1.%s" % ctn))
    assert min(len(r[0]) for r in rows) > 50
    merged = co.merge(texts, ["Cancer", "Non-Cancer"])
    fasta = {n: s.tobytes().decode() for n, s in zip(names, refs)}
    s1 = co.step1(merged, fasta, info_lines=tsvio.STEP1_INFO_LINES)
    cand = [l.split("\t") for l in s1.split("\n") if l and not l.startswith("#") and l.split("\t")[4] != "." and l.split("\t")[5] != "."]
    assert len(cand) > 10
    ed, sr = str(tmp_path / "editing.tsv"), str(tmp_path / "pon.tsv.gz")
    open(ed, "w").write("#c\tp\n" + "".join("%s\t%s\n" % (c[0], c[1]) for c in cand[::7]))
    import gzip
    gzip.open(sr, "wt").write("".join("%s\t%s\n" % (c[0], c[1]) for c in cand[3::9]))
    s2 = co.step2(s1, co.read_posset(ed), {(c[0], int(c[1])) for c in cand[3::9]}, set(), 0, None, 0.01)
    # product
    out = pipeline.run_snv(bam, bct, fa, str(tmp_path / "out"), "S1", editing=ed, pon_sr=sr, engine=engine)
    for ct, ctn in enumerate(("Cancer", "Non-Cancer")):
        assert strip_date(open(out.counts[ctn]).read()) == strip_date(texts[ct])
    assert strip_date(open(out.merged).read()) == strip_date(merged)
    assert strip_date(open(out.step1).read()) == strip_date(neg_zero(s1))
    assert strip_date(open(out.step2).read()) == strip_date(neg_zero(s2))
    # step 3: the host code pinned to the reference's own outputs (tests/test_calling_cpu.py), applied to the ORACLE's step-2 table
    from longsom_amd import calling
    final, unfiltered = calling.step3(neg_zero(s2), 0.05, 0.3, 3, 2, 10000)
    assert open(out.step3).read() == final and open(out.step3_unfiltered).read() == unfiltered
    assert sum(1 for l in final.split("\n") if l and not l.startswith("#")) > 1
    rep = open(out.report).read().split("\n")
    assert rep[0].split("\t")[:4] == ["Total_reads", "Pass_reads", "CB_not_found", "CB_not_matched"]
    assert int(rep[1].split("\t")[0]) == m.n_reads
