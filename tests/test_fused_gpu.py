"""GPU: a load that also makes the first count (lsg_set_count_at_load: pileup.hip k_tm_gather_count builds the store's blocks and adds
their events into the counters in one pass) — the count it hands out, AND every later count over the store it wrote, equal the
events-level CPU oracle bit for bit; loads it cannot serve (more than two cell types, a depth cap that could fire) take the plain
gather and count on request.  Every case is loaded a second time under LSG_STORE_SKIP_WHEN_COUNTED (pileup.hip k_tm_count_direct:
the same count straight from the caller's events, no store written)."""
import numpy as np
import pytest

from longsom_amd import synth
from longsom_amd._lib import CountParams
from tests.support.synth_simple import random_records, random_reference
from tests.test_count_gpu import make_case
from tests.test_fuzz_gpu import draw
from tests.util import phased_records

pytestmark = pytest.mark.gpu


def oracle_rows(rec, lens, refs, ct_of, n_ct, p):
    from oracle import loader
    out, tot = [], 0
    for ct in range(n_ct):
        k, rf, c, ncol = loader.count(rec, lens, refs, ct_of, ct, p.min_bq, p.min_mq, p.min_dp, p.min_cc, p.flag_exclude, p.ignore_orphans)
        out.append((k, rf, c)); tot += ncol
    return out, tot


def fused_vs_oracle(engine, rec, lens, refs, ct_of, n_ct, p, expect_fused=True, recount_params=()):
    """... and all of it once more with the same events laid out tile-phased (LSG_LAYOUT_PHASED: what the device BAM decoder hands over):
    the keys then carry an entry's 128-byte line, and the load that keeps no store fetches every entry as that one line (path 5)"""
    out = _fused_vs_oracle(engine, rec, lens, refs, ct_of, n_ct, p, expect_fused, recount_params)
    ph = phased_records(rec)
    assert _fused_vs_oracle(engine, ph, lens, refs, ct_of, n_ct, p, expect_fused, recount_params, phased=True) == out
    # ... and once more with the caller SAYING that its events are phased (lsg_set_events_layout): the load that keeps no store and sorts keys
    # alone - one whose load filter is the count's own read filter, what the product's loads set - then bins its entries by 128-position windows
    # and fetches every entry as one 256-byte block (path 6); a claim that is wrong is found out and costs a restart
    engine.set_events_layout(engine.LAYOUT_PHASED)
    try:
        assert _fused_vs_oracle(engine, ph, lens, refs, ct_of, n_ct, p, expect_fused, recount_params, phased=True, hinted=True, match_filter=True) == out
        assert _fused_vs_oracle(engine, ph, lens, refs, ct_of, n_ct, p, expect_fused, (), phased=True, hinted=True) == out
        assert _fused_vs_oracle(engine, rec, lens, refs, ct_of, n_ct, p, expect_fused, (), hinted=True, match_filter=True) == out      # (a wrong claim)
    finally:
        engine.set_events_layout(engine.LAYOUT_COMPACT)
    return out


def expected_direct_path(engine, p, phased, hinted, match_filter):
    """lsg_get_layout_info's path of a load that keeps no store: 4 = an entry's events through a descriptor of its own bytes, 5 = tile-phased
    events, one 128-byte line per entry, 6 = entries binned by 128-position windows, one 256-byte block per entry"""
    if not (phased and 1 <= p.min_bq <= 255):
        return 4
    import os
    keys_alone = (match_filter or engine.load_settings()["load_filter"] == (p.min_mq, p.flag_exclude, p.ignore_orphans)) and not os.environ.get("LSG_TEST_KEYS_ONLY_REFUSED") and not os.environ.get("LSG_NO_KEYS_ONLY")
    return 6 if hinted and keys_alone and engine.pileup_window >= 128 else 5


def _fused_vs_oracle(engine, rec, lens, refs, ct_of, n_ct, p, expect_fused, recount_params, phased=False, hinted=False, match_filter=False):
    direct_path = expected_direct_path(engine, p, phased, hinted, match_filter)
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(ct_of, n_ct)
    engine.set_count_at_load(p)
    try:
        engine.load_reads(rec)
    finally:
        engine.set_count_at_load(None)
    assert engine.layout_info()[0] == (3 if expect_fused else 2)
    want, want_cols = oracle_rows(rec, lens, refs, ct_of, n_ct, p)
    for what in ("the count made by the load", "a count over the store that load wrote"):
        rows, cols = engine.pileup_count(p)
        assert cols == want_cols, what
        for ct in range(n_ct):
            k, rf, c = engine.fetch_counts(ct)
            assert rows[ct] == len(want[ct][0]), what
            np.testing.assert_array_equal(k, want[ct][0], err_msg=what); np.testing.assert_array_equal(rf, want[ct][1], err_msg=what)
            np.testing.assert_array_equal(c, want[ct][2], err_msg=what)
    for q in recount_params:                                  # other parameters over the same store
        w2, c2 = oracle_rows(rec, lens, refs, ct_of, n_ct, q)
        rows, cols = engine.pileup_count(q)
        assert cols == c2
        for ct in range(n_ct):
            k, rf, c = engine.fetch_counts(ct)
            np.testing.assert_array_equal(k, w2[ct][0]); np.testing.assert_array_equal(c, w2[ct][2])
    stats_keep = engine.count_stats() if not recount_params else None
    # ... and the same load keeping NO store (lsg_set_store_policy: k_tm_count_direct reads the events where the caller left them)
    saved = engine.load_settings()
    engine.set_count_at_load(p)
    engine.set_store_policy(engine.STORE_SKIP_WHEN_COUNTED)
    if match_filter:
        engine.set_load_filter(p.min_mq, p.flag_exclude, p.ignore_orphans)
    try:
        engine.load_reads(rec)
    finally:
        engine.restore_load_settings(saved)
    assert engine.layout_info()[0] == (direct_path if expect_fused else 2)
    for what in ("the count made by the load that kept no store", "the same count asked for again"):
        rows, cols = engine.pileup_count(p)
        assert cols == want_cols, what
        for ct in range(n_ct):
            k, rf, c = engine.fetch_counts(ct)
            assert rows[ct] == len(want[ct][0]), what
            np.testing.assert_array_equal(k, want[ct][0], err_msg=what); np.testing.assert_array_equal(rf, want[ct][1], err_msg=what)
            np.testing.assert_array_equal(c, want[ct][2], err_msg=what)
    if expect_fused:
        if stats_keep is not None:
            st = engine.count_stats()
            for f in ("n_reads_admitted", "n_segs_admitted", "n_events_admitted") + (("n_entries", "n_units", "n_deep_units") if direct_path != 6 and not match_filter else ()):      # (windows: other entries, other units; a load filter: other jobs)
                assert getattr(st, f) == getattr(stats_keep, f), f
        other = CountParams.longsom_defaults(min_bq=p.min_bq + 1)
        with pytest.raises(RuntimeError, match="kept no store"):
            engine.pileup_count(other)
    return rows, cols


def test_small_and_deep_and_pads(engine):
    p = CountParams.longsom_defaults()
    lens = [5000, 1200, 70]
    rec, refs, ct_of = make_case(1, 3000, lens, 50)
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p, recount_params=[CountParams.longsom_defaults(min_mq=30, min_bq=30)])
    lens = [4000, 2000]                                       # deep tiles: several jobs, slabs, both waves of a job, cuts off a multiple of eight
    rec, refs, ct_of = make_case(3, 30000, lens, 3000, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.9)
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p)
    assert engine.count_stats().n_deep_units > 0
    rec, refs, ct_of = make_case(12, 15000, [3000, 800], 120, hot_regions=[(0, 700, 760)], hot_frac=0.7)      # few barcodes: long runs
    fused_vs_oracle(engine, rec, [3000, 800], refs, ct_of, 2, CountParams.longsom_defaults(min_dp=0, min_cc=0, min_bq=0, min_mq=0))


def test_one_cell_type_and_gates_and_strict_filters(engine):
    lens = [3000, 800]
    rec, refs, ct_of = make_case(12, 9000, lens, 120)
    one = np.zeros(len(ct_of), np.uint8); one[3] = 255
    fused_vs_oracle(engine, rec, lens, refs, one, 1, CountParams.longsom_defaults())
    # the count is stricter than the load filter (none here): admission through the per-read bitmap, in the fused pass too
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(min_mq=60, flag_exclude=0xF04, ignore_orphans=1, min_bq=33))
    engine.set_load_filter(60, 0x704, 1)
    try:
        fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults(), recount_params=[CountParams.longsom_defaults(min_bq=5)])
    finally:
        engine.set_load_filter()


def test_one_barcode_owning_a_tile_is_left_to_the_wide_walk(engine):
    lens = [2500]
    rec, refs, ct_of = make_case(14, 30000, lens, 30, hot_regions=[(0, 700, 760)], hot_frac=0.97, cb_skew=0.9)
    ct_of[0] = 0
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults())


def test_more_than_two_cell_types_and_a_capped_pile_take_the_plain_gather(engine):
    lens = [5000, 700]
    rec, refs, ct_of = make_case(31, 20000, lens, 200, n_ct=4, hot_regions=[(0, 2000, 2100)], hot_frac=0.6)
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 4, CountParams.longsom_defaults(min_dp=3, min_cc=2), expect_fused=False)
    rec, refs, ct_of = make_case(3, 6000, [4000], 300, hot_regions=[(0, 1000, 1100)], hot_frac=0.9)
    engine.set_contigs([4000]); engine.load_reference(0, refs[0]); engine.set_barcodes(ct_of, 2)
    p = CountParams.longsom_defaults(); p.max_depth = 50
    engine.set_count_at_load(p)
    try:
        engine.load_reads(rec)
    finally:
        engine.set_count_at_load(None)
    assert engine.layout_info()[0] == 2                       # the cap could fire: decided by the count itself


@pytest.mark.parametrize("seed", range(14))
def test_random_configuration(engine, seed):
    rng, lens, n_reads, n_cb, n_ct, kw, cp, call = draw(seed)
    refs = [random_reference(rng, L) for L in lens]
    ct_of = rng.integers(0, n_ct, n_cb).astype(np.uint8)
    if n_cb > 4:
        ct_of[rng.random(n_cb) < 0.05] = 255
    rec = random_records(seed + 77, n_reads, lens, n_cb, **kw)
    fused_vs_oracle(engine, rec, lens, refs, ct_of, n_ct, cp, expect_fused=n_ct <= 2)


def test_region_and_device_arrays(engine):
    """a rank's share: the region is set before the load (tiles outside it are gathered, not counted); arrays generated in HBM"""
    m = synth.named("C1", n_reads=20_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    p = CountParams.longsom_defaults()
    engine.set_region()
    engine.synth_reads(m)
    whole = engine.pileup_count(p)
    ref_rows = [engine.fetch_counts(ct) for ct in range(2)]
    mid = int(m.contig_len[0]) // 2 // 64 * 64
    for policy, path in ((engine.STORE_SKIP_WHEN_COUNTED, 4), (engine.STORE_KEEP, 3)):
        parts = []
        for lo, hi in (((0, 0), (0, mid)), ((0, mid), (len(m.contig_len), 0))):
            engine.set_region(lo[0], lo[1], hi[0], hi[1])
            engine.set_count_at_load(p)
            engine.set_store_policy(policy)
            try:
                engine.synth_reads(m)
            finally:
                engine.set_count_at_load(None)
                engine.set_store_policy(engine.STORE_KEEP)
            assert engine.layout_info()[0] == path
            rows, cols = engine.pileup_count(p)
            parts.append((rows, cols, [engine.fetch_counts(ct) for ct in range(2)]))
        assert parts[0][1] + parts[1][1] == whole[1]
        for ct in range(2):
            for j in range(3):
                np.testing.assert_array_equal(np.concatenate([parts[0][2][ct][j], parts[1][2][ct][j]]), ref_rows[ct][j])
    engine.set_region()
    rows, cols = engine.pileup_count(p)                       # the whole genome over the store the second (regional) load wrote
    assert (rows, cols) == whole


def test_entries_cut_at_pileup_window_edges_count_the_same(engine):
    """narrow pileup windows (an edge inside most tiles): the entries are cut at the edges, the counts are not"""
    lens = [4000, 2000]
    rec, refs, ct_of = make_case(3, 30000, lens, 3000, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.9)
    engine.set_pileup_window(100)
    try:
        fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults())
        n_cut = engine.store_shape()[0]
    finally:
        engine.set_pileup_window(50000)
    fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, CountParams.longsom_defaults())
    assert n_cut > engine.store_shape()[0]                        # more entries, same rows


def test_loads_that_sort_keys_alone(engine, monkeypatch, capfd):
    """A load that keeps no store and is counted under its own read filters carries keys alone through the scatter and the sort
    (store.hip build_store, keys_only; pileup.hip k_tm_count_direct<true>): deep tiles cut into jobs, a tile left to the wide walk,
    and a load whose count from keys alone is refused (the test hook) and that is made again with values."""
    monkeypatch.setenv("LSG_TIMING", "1")
    p = CountParams.longsom_defaults()
    engine.set_load_filter(p.min_mq, p.flag_exclude, p.ignore_orphans)           # every stored read is admitted by the count
    try:
        lens = [4000, 2000]
        rec, refs, ct_of = make_case(3, 30000, lens, 3000, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.9)
        capfd.readouterr()
        fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p)
        assert "keys alone through the scatter" in capfd.readouterr().err
        assert engine.count_stats().n_deep_units > 0
        lens = [2500]
        rec, refs, ct_of = make_case(14, 30000, lens, 30, hot_regions=[(0, 700, 760)], hot_frac=0.97, cb_skew=0.9)
        ct_of[0] = 0
        fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p)
        assert "keys alone through the scatter" in capfd.readouterr().err
        one = np.zeros(len(ct_of), np.uint8); one[3] = 255
        fused_vs_oracle(engine, rec, lens, refs, one, 1, p)
        monkeypatch.setenv("LSG_TEST_KEYS_ONLY_REFUSED", "1")
        lens = [5000, 1200, 70]
        rec, refs, ct_of = make_case(1, 3000, lens, 50)
        capfd.readouterr()
        fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p)
        assert "loading again with values" in capfd.readouterr().err
    finally:
        engine.set_load_filter()


def test_library_paths_behind_the_switches_still_count_the_same(engine, monkeypatch):
    """The tiles' tables as library calls (LSG_NO_TILE_TABLES), the 256-thread sort of a small load (LSG_SORT_BIG_BLOCKS=0) and a load that
    carries values although it could sort keys alone (LSG_NO_KEYS_ONLY): the paths the defaults replaced, against the oracle."""
    p = CountParams.longsom_defaults()
    lens = [4000, 2000]
    rec, refs, ct_of = make_case(3, 30000, lens, 3000, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.9)
    engine.set_load_filter(p.min_mq, p.flag_exclude, p.ignore_orphans)
    try:
        for name, val in (("LSG_NO_TILE_TABLES", "1"), ("LSG_SORT_BIG_BLOCKS", "0"), ("LSG_NO_KEYS_ONLY", "1")):
            monkeypatch.setenv(name, val)
            fused_vs_oracle(engine, rec, lens, refs, ct_of, 2, p)
            monkeypatch.delenv(name)
    finally:
        engine.set_load_filter()
