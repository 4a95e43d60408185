"""Known-answer tests of the pileup stage (hand-derived expectations, tests/kat_pileup_cases.py).

CPU (not gpu): the BAM-level oracle (oracle/plp_oracle.c) and the decoder + events-level oracle both reproduce
the hand-derived rows.  GPU: decoder + HIP kernels reproduce them too."""
import json
import os

import numpy as np
import pytest

from longsom_amd import hostio
from tests.support import bamwrite
from longsom_amd._lib import CountParams
from oracle import loader
from tests import kat_pileup_cases as K

REF = np.frombuffer(K.REF.encode(), dtype=np.uint8)
BARCODES = [b for b, _ in K.BARCODES]
CT_OF = np.array([0 if t == "Cancer" else 1 for _, t in K.BARCODES], np.uint8)


def make_bam(tmp_path, name):
    reads = sorted(K.CASES[name]["reads"], key=lambda r: r["pos"])
    p = str(tmp_path / (name + ".bam"))
    bamwrite.write_bam(p, [K.CONTIG], reads)
    return p


def check(keys, counts, spec, what):
    want = K.expected_rows(spec)
    got = {int(k & 0xFFFFFFFF): c for k, c in zip(keys.tolist(), counts)}
    assert sorted(got) == sorted(want), "%s: emitted positions %s, expected %s" % (what, sorted(p + 1 for p in got), sorted(p + 1 for p in want))
    for pos, row in want.items():
        g = got[pos]
        assert [int(g[i]) for i in K.PRINTED] == [row[i] for i in K.PRINTED], "%s pos %d: got %s want %s" % (what, pos + 1, list(map(int, g)), row)


@pytest.mark.parametrize("name", sorted(K.CASES))
def test_oracles_reproduce_hand_derived_rows(tmp_path, name):
    case = K.CASES[name]
    bam = make_bam(tmp_path, name)
    p = case["params"]
    for ct, key in ((0, "cancer"), (1, "noncancer")):
        spec = case.get(key, {} if key == "noncancer" else None)
        k, r, c = loader.plp_count(bam, BARCODES, CT_OF, ct, [K.CONTIG[1]], [REF], **p)
        check(k, c, spec, "plp_oracle/" + key)
        dec = hostio.decode_bam(bam, BARCODES, min_mapq=p["min_mq"])
        assert dec.contig_names == [K.CONTIG[0]] and list(dec.contig_len) == [K.CONTIG[1]]
        k2, r2, c2, _ = loader.count(dec.records, [K.CONTIG[1]], [REF], CT_OF, ct, p["min_bq"], p["min_mq"], p["min_dp"], p["min_cc"])
        check(k2, c2, spec, "decoder+count_oracle/" + key)
        np.testing.assert_array_equal(k, k2); np.testing.assert_array_equal(c, c2)


def test_split_bam_report_counters(tmp_path):
    dec = hostio.decode_bam(make_bam(tmp_path, "barcodes"), BARCODES, min_mapq=60)
    assert dec.report == {"Total_reads": 4, "Pass_reads": 2, "CB_not_found": 1, "CB_not_matched": 1}
    dec = hostio.decode_bam(make_bam(tmp_path, "mapq"), BARCODES, min_mapq=60)
    assert dec.report == {"Total_reads": 2, "Pass_reads": 1, "CB_not_found": 0, "CB_not_matched": 0, "MAPQ": 1}


def test_kat_json_is_current():
    """tests/golden/kat_pileup.json is the data form of the cases (for readers of the fixtures)."""
    path = os.path.join(os.path.dirname(__file__), "golden", "kat_pileup.json")
    data = {n: {"reads": c["reads"], "params": c["params"], "cancer": {str(k): v for k, v in K.expected_rows(c["cancer"]).items()},
                "noncancer": {str(k): v for k, v in K.expected_rows(c.get("noncancer", {})).items()}} for n, c in K.CASES.items()}
    if not os.path.exists(path):
        json.dump({"contig": K.CONTIG, "reference": K.REF, "barcodes": K.BARCODES, "cases": data}, open(path, "w"), indent=1, sort_keys=True)
    on_disk = json.load(open(path))
    assert json.loads(json.dumps(data)) == on_disk["cases"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(K.CASES))
def test_gpu_reproduces_hand_derived_rows(tmp_path, engine, name):
    case = K.CASES[name]
    dec = hostio.decode_bam(make_bam(tmp_path, name), BARCODES, min_mapq=case["params"]["min_mq"])
    engine.set_contigs([K.CONTIG[1]]); engine.load_reference(0, REF); engine.set_barcodes(CT_OF, 2); engine.set_region()
    engine.load_reads(dec.records)
    p = case["params"]
    engine.pileup_count(CountParams.longsom_defaults(**p))
    for ct, key in ((0, "cancer"), (1, "noncancer")):
        k, r, c = engine.fetch_counts(ct)
        check(k, c, case.get(key, {}), "gpu/" + key)
