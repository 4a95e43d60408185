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
