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
