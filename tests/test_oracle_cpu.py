"""CPU: the calling oracle (oracle/calling_oracle.py) pinned to the goldens the reference's own code produced."""
import json
import os

from longsom_amd import tsvio
from oracle import calling_oracle as co

G = os.path.join(os.path.dirname(__file__), "golden")


def rd(name):
    return open(os.path.join(G, name)).read()


def strip_date(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("##fileDate="))


def test_merge_oracle():
    got = co.merge([rd("counts.sample.Cancer.tsv"), rd("counts.sample.Non-Cancer.tsv")], ["Cancer", "Non-Cancer"])
    assert strip_date(got) == strip_date(rd("merged.tsv"))


def test_step1_oracle():
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    fasta = {n: s.tobytes().decode() for n, s in zip(names, seqs)}
    got = co.step1(rd("merged.tsv"), fasta, info_lines=tsvio.STEP1_INFO_LINES)
    assert got == rd("sample.calling.step1.tsv")


def test_step2_oracle():
    ed, sr, lr = (co.read_posset(os.path.join(G, "calling.%s.tsv" % k)) for k in ("editing", "pon_SR", "pon_LR"))
    af = json.load(open(os.path.join(G, "calling.gnomad_af.json")))
    got = co.step2(rd("sample.calling.step1.tsv"), ed, sr, lr, 0, af, 0.01)
    assert got == rd("sample.calling.step2.tsv")
    got = co.step2(rd("sample.calling.step1.tsv"), ed, sr, co.read_posset(""), 150, af, 0.01)
    assert got == rd("sample.dist150.calling.step2.tsv")


def test_region_parallel_count_oracle_equals_the_single_threaded_one():
    """lso_count_mt (regions over threads, the form that writes the scale pins and the all-cores CPU baseline) == lso_count"""
    import numpy as np
    from longsom_amd import hostio, synth
    from oracle import loader
    m = synth.named("C1", n_reads=6000, n_genes=40)
    rec = hostio.synth_records(m)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    for ct in range(2):
        a = loader.count(rec, m.contig_len, refs, m.celltype_of, ct)
        for threads, w in ((3, 512), (8, 64), (2, 100000)):
            b = loader.count(rec, m.contig_len, refs, m.celltype_of, ct, threads=threads, region_w=w)
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3] and len(a[0]) > 100
