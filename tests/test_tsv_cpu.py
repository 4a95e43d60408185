"""CPU: text writers/parsers against the golden files produced by the reference (tests/golden, tools/make_goldens.py)."""
import os

import numpy as np

from longsom_amd import tsvio

G = os.path.join(os.path.dirname(__file__), "golden")
CONTIGS = ["chr1", "chr10", "chr2", "chrM"]


def strip_date(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("##fileDate="))


def test_counts_roundtrip():
    for ct in ("Cancer", "Non-Cancer"):
        path = os.path.join(G, "counts.sample.%s.tsv" % ct)
        keys, refs, counts, sid = tsvio.parse_counts_tsv(path, CONTIGS)
        assert sid == "sample.%s" % ct and len(keys) > 500 and np.all(np.diff(keys) > 0)
        text = tsvio.format_counts_tsv(keys, refs, counts, CONTIGS, sid)
        assert strip_date(text) == strip_date(open(path).read())


def test_merge_matches_reference_golden():
    per_ct = [tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.%s.tsv" % ct), CONTIGS)[:3] for ct in ("Cancer", "Non-Cancer")]
    text = tsvio.format_merged_tsv(per_ct, CONTIGS, ["Cancer", "Non-Cancer"])
    assert strip_date(text) == strip_date(open(os.path.join(G, "merged.tsv")).read())
