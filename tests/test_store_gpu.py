"""GPU: the tile store lsg_load_reads builds is independent of count parameters, barcode table and region — every count over it
equals the CPU oracle — and it is the only resident copy of the events."""
import numpy as np
import pytest

from longsom_amd import synth
from longsom_amd._lib import CountParams
from tests.test_count_gpu import make_case, run_both

pytestmark = pytest.mark.gpu


def load(engine, rec, lens, refs, ct_of, n_ct):
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(ct_of, n_ct)
    engine.load_reads(rec)


def check(engine, rec, lens, refs, ct_of, n_ct, p):
    """the resident store counted under p == the events-level oracle"""
    from oracle import loader
    n_rows, n_cols = engine.pileup_count(p)
    tot = 0
    out = []
    for ct in range(n_ct):
        k, rf, c = engine.fetch_counts(ct)
        ok, orf, oc, ocols = loader.count(rec, lens, refs, ct_of, ct, p.min_bq, p.min_mq, p.min_dp, p.min_cc, p.flag_exclude, p.ignore_orphans)
        tot += ocols
        np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(rf, orf); np.testing.assert_array_equal(c, oc)
        out.append((k, rf, c))
    assert tot == n_cols
    return out


def test_one_store_serves_every_filter_table_and_reads(engine):
    """read filters and the barcode table are resolved per count: one load, many counts, no rebuild"""
    lens = [3000, 800]
    rec, refs, ct_of = make_case(12, 15000, lens, 120, hot_regions=[(0, 700, 760)], hot_frac=0.7)
    load(engine, rec, lens, refs, ct_of, 2)
    built = engine.layout_info()[1]
    assert built > 0 and engine.layout_info()[2] > 0
    for p in (CountParams.longsom_defaults(), CountParams.longsom_defaults(min_mq=0, min_bq=0, min_dp=0, min_cc=0), CountParams.longsom_defaults(min_mq=30),
              CountParams.longsom_defaults(flag_exclude=0x704), CountParams.longsom_defaults(ignore_orphans=0, min_bq=35)):
        check(engine, rec, lens, refs, ct_of, 2, p)
    # re-annotation: another table over the same reads (the store stays, the classes change), then one and three cell types
    ct2 = ct_of.copy(); ct2[::3] = 1 - np.minimum(ct2[::3], 1); ct2[5] = 255
    p = CountParams.longsom_defaults()
    engine.set_barcodes(ct2, 2)
    check(engine, rec, lens, refs, ct2, 2, p)
    ct3 = (np.arange(len(ct_of)) % 3).astype(np.uint8); ct3[7] = 255
    engine.set_barcodes(ct3, 3)
    check(engine, rec, lens, refs, ct3, 3, p)
    ct1 = np.zeros(len(ct_of), np.uint8)
    engine.set_barcodes(ct1, 1)
    check(engine, rec, lens, refs, ct1, 1, p)
    assert engine.layout_info()[1] == built                      # nothing was rebuilt
    # other reads
    rec2, _, _ = make_case(13, 9000, lens, 120)
    engine.set_barcodes(ct_of, 2)
    engine.load_reads(rec2)
    check(engine, rec2, lens, refs, ct_of, 2, p)


def test_four_cell_types_take_two_passes(engine):
    lens = [5000, 700]
    rec, refs, ct_of = make_case(31, 20000, lens, 200, n_ct=4, hot_regions=[(0, 2000, 2100)], hot_frac=0.6)
    rows, cols = run_both(engine, rec, lens, refs, ct_of, 4, CountParams.longsom_defaults(min_dp=3, min_cc=2))
    assert all(r > 0 for r in rows)
    assert engine.count_stats().n_deep_units > 0                 # multi-job tiles: slabs of all four cell types


def test_one_barcode_owning_a_tile(engine):
    """a single barcode's run of more entries than the packed planes' fields hold cannot be cut: the wide walk takes that job,
    in a single-job tile and in a multi-job one"""
    lens = [2500]
    rec, refs, ct_of = make_case(14, 30000, lens, 30, hot_regions=[(0, 700, 760)], hot_frac=0.97, cb_skew=0.9)
    ct_of[0] = 0
    run_both(engine, rec, lens, refs, ct_of, 2)
    rec, refs, ct_of = make_case(15, 9000, lens, 3, hot_regions=[(0, 100, 140)], hot_frac=0.99, cb_skew=0.99)
    ct_of[:] = [0, 1, 0]
    run_both(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_cc=1))


def test_region_counts_add_up(engine):
    lens = [6000, 1500]
    rec, refs, ct_of = make_case(15, 20000, lens, 150, hot_regions=[(0, 3000, 3100)], hot_frac=0.5)
    load(engine, rec, lens, refs, ct_of, 2)
    p = CountParams.longsom_defaults()
    engine.set_region()
    whole = check(engine, rec, lens, refs, ct_of, 2, p)
    parts = []
    for lo, hi in (((0, 0), (0, 2944)), ((0, 2944), (1, 640)), ((1, 640), (2, 0))):
        engine.set_region(lo[0], lo[1], hi[0], hi[1])
        engine.pileup_count(p)
        parts.append([engine.fetch_counts(ct) for ct in range(2)])
    engine.set_region()
    for ct in range(2):
        k = np.concatenate([pr[ct][0] for pr in parts]); c = np.concatenate([pr[ct][2] for pr in parts])
        np.testing.assert_array_equal(k, whole[ct][0]); np.testing.assert_array_equal(c, whole[ct][2])


def test_device_arrays_load_like_host_arrays(engine):
    """lsg_load_reads of device-resident arrays (what lsg_synth_generate hands out) == the same model loaded by lsg_synth_reads, and
    the events are not kept: the store and the small per-read / per-segment arrays are all that stays"""
    m = synth.named("C1", n_reads=20_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2); engine.set_region()
    engine.synth_reads(m)
    rows, cols = engine.pileup_count()
    a = [engine.fetch_counts(ct) for ct in range(2)]
    g = engine.synth_generate(m)
    assert g.on_device == 1 and g.n_reads == 20_000
    for _ in range(2):                                           # loaded twice: a fresh store each time
        engine.load_reads_struct(g)
        assert engine.pileup_count() == (rows, cols)
    b = [engine.fetch_counts(ct) for ct in range(2)]
    for (k1, r1, c1), (k2, r2, c2) in zip(a, b):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(c1, c2)
    ms = engine.build_times()
    assert len(ms) == 4 and all(x >= 0 for x in ms) and sum(ms) > 0
    entries, blocks, events = engine.store_shape()
    assert events <= g.n_events and entries > 0 and blocks * 8 >= entries


def test_resident_bytes_of_a_load():
    """what a load leaves resident (a fresh handle: buffers are grow-only): the store, the per-entry words, the small per-read and
    per-segment arrays and the build's cached temporaries — no second copy of the events"""
    from longsom_amd.engine import Engine
    m = synth.named("C2", n_reads=400_000)
    with Engine(0) as eng:
        eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2)
        ref_bytes = int(np.sum(m.contig_len))
        eng.synth_reads(m)
        n_reads, n_segs, n_events = eng.reads_shape()
        entries, blocks, events = eng.store_shape()
        store_bytes = eng.layout_info()[2]
        # blocks (1 KB per 8 entries) + 17 B of words per entry + 8 B of cached gather sources + 44 B of sort temporaries per entry (two sorts side by side: 12 B of scratch more) + tile tables
        # (grow-only buffers reserve 1/16 more than asked); the tile tables include the plan's tile-level half (32 B per tile), which the load
        # makes beside its gather when the number of cell types is known
        bound = 1.07 * (blocks * 1024 + entries * (17 + 8 + 44) * 1.1 + (24 + 36) * (ref_bytes // 64) + 27 * n_reads + 20 * n_segs) + (64 << 20)
        assert store_bytes < bound, (store_bytes, bound)
        with pytest.raises(Exception, match="not kept"):
            eng.reads_to_host()


def test_unsorted_barcode_ids_and_sparse_ids(engine):
    """barcode ids far apart (the sort covers the largest id present) and entries whose barcode is not in the table"""
    lens = [4000]
    rec, refs, ct_of = make_case(41, 6000, lens, 3000)
    ct_of[:] = 255
    ct_of[[1, 17, 1024, 2999]] = [0, 1, 0, 1]
    run_both(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_dp=1, min_cc=1))


def test_load_filter_drops_reads_like_splitbam(engine):
    """lsg_set_load_filter: reads that fail it never reach the store (SplitBamCellTypes.py:110-113 filters the BAM before
    BaseCellCounter); counts at least as strict equal the oracle, a looser count is refused"""
    from longsom_amd._lib import LsgError
    lens = [3000, 800]
    rec, refs, ct_of = make_case(51, 12000, lens, 100, hot_regions=[(0, 700, 760)], hot_frac=0.5)
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(ct_of, 2)
    engine.set_load_filter()
    engine.load_reads(rec)
    full = engine.store_shape()
    p = CountParams.longsom_defaults()
    try:
        engine.set_load_filter(p.min_mq, p.flag_exclude, p.ignore_orphans)
        engine.load_reads(rec)
        small = engine.store_shape()
        assert small[0] < full[0] and small[2] < full[2]
        check(engine, rec, lens, refs, ct_of, 2, p)
        check(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_mq=61, min_bq=25))
        with pytest.raises(LsgError, match="load filter"):
            engine.pileup_count(CountParams.longsom_defaults(min_mq=30))
        with pytest.raises(LsgError, match="load filter"):
            engine.pileup_count(CountParams.longsom_defaults(flag_exclude=0x704))
    finally:
        engine.set_load_filter()
    engine.load_reads(rec)
    check(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_mq=30))


def test_unload_gives_the_memory_back_and_the_handle_stays_usable():
    """lsg_unload_reads: the load's device memory returns (within the allocator's granularity), counting without reads is an error,
    the next load counts as the first one did"""
    import ctypes as C
    from longsom_amd import synth
    from longsom_amd._lib import LsgError
    from longsom_amd.engine import Engine
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    m = synth.named("C1", n_reads=200_000)
    with Engine(0) as eng:
        eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2)
        free0 = free_bytes()
        eng.synth_reads(m)
        first = eng.pileup_count()
        eng.call_step1()
        used = free0 - free_bytes()
        assert used > 50 << 20
        eng.unload_reads()
        assert free0 - free_bytes() < used // 8
        with pytest.raises(LsgError):
            eng.pileup_count()
        eng.synth_reads(m)
        assert eng.pileup_count() == first


def test_a_load_without_reads_counts_nothing(engine):
    """what a rank of a sharded run does when no alignment falls into its region (pipeline._run_snv_regions): an empty load, a count, a
    call, fetches — all of them empty, none of them an error"""
    from longsom_amd import hostio
    m = synth.named("C1", n_reads=1000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.load_reads(hostio.ReadRecords.empty())
    rows, cols = engine.pileup_count()
    assert rows == [0, 0] and cols == 0
    assert engine.call_step1() == (0, 0)
    assert all(len(x) == 0 for x in engine.fetch_counts(0)) and len(engine.fetch_calls()) == 0
