"""CPU: step 3 (host side, pandas) against the reference's golden outputs."""
import os

import pytest

from longsom_amd import calling

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
