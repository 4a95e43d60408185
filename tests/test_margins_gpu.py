"""GPU: exactness margin of the printed p-values.  The device's fp64 tails differ from scipy's by ~1e-13; a
printed 4-decimal value can only differ where p lies that close to a rounding tie.  Over every distinct (k, n) the candidates of a
C2 sample were tested with: no p is closer than 1e-9 to a tie unless scipy itself, asked for that very pair, prints the same
digits (tools/p_margins.py does the full-size audit; its result is kept under profiles/)."""
import sys, os

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from longsom_amd import synth
from longsom_amd._lib import CallParams

pytestmark = pytest.mark.gpu


def test_no_printed_p_value_sits_on_a_rounding_tie(engine):
    import p_margins
    m = synth.named("C2", n_reads=600_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2); engine.set_region()
    engine.synth_reads(m)
    engine.pileup_count(); engine.call_step1()
    calls = engine.fetch_calls(candidates_only=True)
    per_ct = [engine.fetch_counts(ct) for ct in range(2)]
    res = p_margins.audit(engine, calls, per_ct, CallParams.longsom_defaults(), near=1e-6)
    for name in ("reads", "cells"):
        r = res[name]
        assert r["pairs"] > 1000
        assert r["differ_from_scipy"] == [], r["differ_from_scipy"][:3]
        assert r["min_distance"] > 1e-12 or r["within_near"] > 0
