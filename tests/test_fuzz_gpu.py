"""GPU: seeded random configurations through the whole device chain, each compared with the CPU oracles: count rows bit-exact
against oracle/count_oracle.c, then merge + step-1 call of THOSE rows against oracle/calling_oracle.py (byte-identical tables).
Shapes the fixed cases do not reach: contigs shorter than a tile or not a multiple of 64, one barcode, 1-4 cell types, gates
off, odd quality / MAPQ thresholds, hot spots with skewed barcodes, call thresholds other than LongSom's."""
import numpy as np
import pytest

from tests.util import neg_zero

from longsom_amd import tsvio
from longsom_amd._lib import CallParams, CountParams
from tests.support.synth_simple import random_records, random_reference
from tests.test_count_gpu import run_both

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(9000 + seed)
    n_contigs = int(rng.integers(1, 5))
    lens = [int(rng.choice([37, 64, 65, 130, 700, 1500, 4099, 9000])) for _ in range(n_contigs)]
    if max(lens) < 130:
        lens[0] = 1500
    n_reads = int(rng.choice([3, 60, 500, 2500, 6000]))
    n_cb = int(rng.choice([1, 2, 17, 120, 300]))
    n_ct = int(rng.integers(1, 5))
    kw = {}
    big = int(np.argmax(lens))
    if rng.random() < 0.5 and lens[big] > 400:
        s = int(rng.integers(0, lens[big] - 200))
        kw = dict(hot_regions=[(big, s, s + int(rng.integers(5, 150)))], hot_frac=float(rng.choice([0.5, 0.95])))
    if rng.random() < 0.3:
        kw["cb_skew"] = float(rng.choice([0.3, 0.8]))
    cp = CountParams.longsom_defaults(min_bq=int(rng.choice([0, 10, 20, 30, 41])), min_mq=int(rng.choice([0, 30, 60])),
                                      min_dp=int(rng.choice([0, 1, 5])), min_cc=int(rng.choice([0, 1, 5])),
                                      ignore_orphans=int(rng.integers(0, 2)))
    call = dict(min_ac_cells=int(rng.choice([1, 2, 3])), min_ac_reads=int(rng.choice([1, 3, 5])), min_cells=int(rng.choice([1, 5])),
                min_cell_types=int(rng.integers(1, n_ct + 1)), alpha1=float(rng.choice([0.21356677091082193, 0.5])),
                beta2=float(rng.choice([162.03696139428595, 40.0])))
    return rng, lens, n_reads, n_cb, n_ct, kw, cp, call


@pytest.mark.parametrize("seed", range(14))
def test_random_configuration(engine, seed):
    from oracle import calling_oracle
    rng, lens, n_reads, n_cb, n_ct, kw, cp, call = draw(seed)
    refs = [random_reference(rng, L) for L in lens]
    ct_of = rng.integers(0, n_ct, n_cb).astype(np.uint8)
    if n_cb > 4:
        ct_of[rng.random(n_cb) < 0.05] = 255
    rec = random_records(seed + 77, n_reads, lens, n_cb, **kw)
    run_both(engine, rec, lens, refs, ct_of, n_ct, cp)                       # count rows == C oracle, bit for bit

    # merge + step 1 of the rows just counted
    per_ct = [engine.fetch_counts(ct) for ct in range(n_ct)]
    names = ["ctg%d" % i for i in range(len(lens))]
    ct_names = ["T%d" % i for i in range(n_ct)]
    params = CallParams.longsom_defaults(min_cov=5, **call)
    n_sites, _ = engine.call_step1(params)
    calls = engine.fetch_calls()
    assert len(calls) == n_sites
    merged = tsvio.format_merged_tsv(per_ct, names, ct_names)
    header = [l + "\n" for l in merged.split("\n") if l.startswith("##")]
    got = tsvio.format_step1_tsv(calls, per_ct, names, ct_names, header)
    fasta = {n: r.tobytes().decode() for n, r in zip(names, refs)}
    want = neg_zero(calling_oracle.step1(merged, fasta, alpha1=call["alpha1"], beta2=call["beta2"], min_ac_cells=call["min_ac_cells"],
                                         min_ac_reads=call["min_ac_reads"], min_cells=call["min_cells"], min_cell_types=call["min_cell_types"],
                                         info_lines=tsvio.STEP1_INFO_LINES))
    g = [l for l in got.split("\n") if l and not l.startswith("##fileDate=")]
    w = [l for l in want.split("\n") if l and not l.startswith("##fileDate=")]
    bad = [(a, b) for a, b in zip(g, w) if a != b]
    assert len(g) == len(w) and not bad, "first mismatch:\n%s\n%s" % (bad[0] if bad else (len(g), len(w)))
