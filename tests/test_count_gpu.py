"""GPU parity of the pileup count kernels against the events-level oracle (bit-exact, integers)."""
import numpy as np
import pytest

from longsom_amd._lib import CountParams
from tests.support.synth_simple import random_records, random_reference

pytestmark = pytest.mark.gpu


def run_both(engine, rec, contig_lens, refs, celltype_of, n_ct, params=None):
    from oracle import loader
    params = params or CountParams.longsom_defaults()
    engine.set_contigs(contig_lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(celltype_of, n_ct)
    engine.load_reads(rec)
    n_rows, n_cols = engine.pileup_count(params)
    tot_cols = 0
    for ct in range(n_ct):
        k, rf, c = engine.fetch_counts(ct)
        ok, orf, oc, ocols = loader.count(rec, contig_lens, refs, celltype_of, ct, params.min_bq, params.min_mq,
                                          params.min_dp, params.min_cc, params.flag_exclude, params.ignore_orphans)
        tot_cols += ocols
        assert n_rows[ct] == len(ok), f"ct {ct}: rows {n_rows[ct]} vs oracle {len(ok)}"
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_array_equal(rf, orf)
        np.testing.assert_array_equal(c, oc)
    assert n_cols == tot_cols
    return n_rows, n_cols


def make_case(seed, n_reads, contig_lens, n_cb, n_ct=2, **kw):
    rng = np.random.default_rng(seed + 1000)
    refs = [random_reference(rng, int(L)) for L in contig_lens]
    celltype_of = rng.integers(0, n_ct, n_cb).astype(np.uint8)
    celltype_of[rng.random(n_cb) < 0.05] = 255
    rec = random_records(seed, n_reads, contig_lens, n_cb, **kw)
    return rec, refs, celltype_of


def test_small_random(engine):
    lens = [5000, 1200, 70]
    rec, refs, ct_of = make_case(1, 3000, lens, 50)
    rows, cols = run_both(engine, rec, lens, refs, ct_of, 2)
    assert sum(rows) > 0


def test_gates_off(engine):
    lens = [3000]
    rec, refs, ct_of = make_case(2, 500, lens, 20)
    run_both(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_dp=0, min_cc=0, min_bq=0, min_mq=0))


def test_deep_units(engine):
    """hot region deep enough for the workgroup kernel with several staged passes."""
    lens = [4000, 2000]
    rec, refs, ct_of = make_case(3, 30000, lens, 3000, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.9)
    run_both(engine, rec, lens, refs, ct_of, 2)
    assert engine.count_stats().n_deep_units > 0


def test_deep_single_barcode_stream_mode(engine):
    """one barcode owning thousands of entries of a tile forces the deep kernel's stream mode."""
    lens = [2500]
    rec, refs, ct_of = make_case(4, 12000, lens, 40, hot_regions=[(0, 700, 760)], hot_frac=0.95, cb_skew=0.6)
    ct_of[0] = 0
    run_both(engine, rec, lens, refs, ct_of, 2)


def test_three_cell_types_and_min_mq(engine):
    lens = [6000, 300]
    rec, refs, ct_of = make_case(5, 4000, lens, 80, n_ct=3)
    run_both(engine, rec, lens, refs, ct_of, 3, CountParams.longsom_defaults(min_mq=30, min_bq=10, min_dp=3, min_cc=2))


def test_more_cell_types_than_the_run_before():
    """a handle that counted a large two-cell-type sample then gets a small one with three: the third cell type's row planes
    must exist although the row capacity did not have to grow (own handle: the shared one has seen four cell types already)."""
    from longsom_amd.engine import Engine
    eng = Engine(0)
    lens = [6000, 300]
    rec, refs, ct_of = make_case(7, 6000, lens, 80, n_ct=2)
    run_both(eng, rec, lens, refs, ct_of, 2)
    rec, refs, ct_of = make_case(5, 600, lens, 80, n_ct=3)
    run_both(eng, rec, lens, refs, ct_of, 3, CountParams.longsom_defaults(min_dp=3, min_cc=2))
    eng.close()


def test_empty_and_tiny(engine):
    lens = [1000]
    rec, refs, ct_of = make_case(6, 1, lens, 5)
    run_both(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_dp=1, min_cc=1))
    rec0 = rec.subset(np.zeros(rec.n_reads, bool))
    engine.load_reads(rec0)
    rows, cols = engine.pileup_count()
    assert rows == [0, 0] and cols == 0


def test_read_order_invariance(engine):
    lens = [5000]
    rec, refs, ct_of = make_case(7, 2000, lens, 64)
    engine.set_contigs(lens); engine.load_reference(0, refs[0]); engine.set_barcodes(ct_of, 2)
    engine.load_reads(rec); engine.pileup_count()
    a = [engine.fetch_counts(ct) for ct in range(2)]
    perm = np.random.default_rng(0).permutation(rec.n_reads)
    # permute reads: rebuild records in permuted order
    order = np.argsort(np.argsort(perm))
    from longsom_amd.engine import ReadRecords
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    rec2 = ReadRecords(rec.read_tid[perm], rec.read_pos[perm], rec.read_flag[perm], rec.read_mapq[perm], rec.read_cb[perm],
                       inv[rec.seg_read].astype(np.uint32), rec.seg_start, rec.seg_len, rec.seg_ev_off, rec.events)
    engine.load_reads(rec2); engine.pileup_count()
    b = [engine.fetch_counts(ct) for ct in range(2)]
    for (k1, r1, c1), (k2, r2, c2) in zip(a, b):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(c1, c2)


def test_bad_event_range_is_an_error(engine):
    """a segment pointing outside the events array is refused at load time (no out-of-bounds read on the device)"""
    import dataclasses
    lens = np.array([3000], np.int64)
    rec, refs, celltype_of = make_case(5, 60, lens, 8)
    engine.set_contigs(lens)
    engine.load_reference(0, refs[0])
    engine.set_barcodes(celltype_of, 2)
    off = rec.seg_ev_off.copy()
    off[-1] = rec.n_events
    with pytest.raises(RuntimeError, match="outside the events array"):
        engine.load_reads(dataclasses.replace(rec, seg_ev_off=off))
    assert engine.reads_shape() == (0, 0, 0)                      # a refused load leaves no reads behind
    sr = rec.seg_read.copy()
    sr[0] = rec.n_reads                                           # a segment owned by a read that does not exist
    with pytest.raises(RuntimeError, match="read index"):
        engine.load_reads(dataclasses.replace(rec, seg_read=sr))
    run_both(engine, rec, lens, refs, celltype_of, 2)


def test_few_barcodes_with_long_runs(engine):
    """many entries per barcode in the deep tiles: long runs, multi-job tiles cut at run starts"""
    lens = np.array([4000], np.int64)
    rec, refs, celltype_of = make_case(21, 9000, lens, 300, hot_regions=[(0, 500, 900)], hot_frac=0.9)
    run_both(engine, rec, lens, refs, celltype_of, 2)
    rec, refs, celltype_of = make_case(22, 7000, lens, 6, hot_regions=[(0, 500, 900)], hot_frac=0.9, cb_skew=0.7)
    run_both(engine, rec, lens, refs, celltype_of, 2)
