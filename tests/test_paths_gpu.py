"""GPU: the three forms of the count (scatter + sort per count, tile index, tile-major store) write the same rows, and the
per-load structures follow the reads, the read filters and the barcode table."""
import os

import numpy as np
import pytest

from longsom_amd._lib import CountParams
from tests.test_count_gpu import make_case

pytestmark = pytest.mark.gpu


def load(engine, rec, lens, refs, ct_of, n_ct):
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(ct_of, n_ct)
    engine.load_reads(rec)


def rows_of(engine, n_ct, params):
    n_rows, n_cols = engine.pileup_count(params)
    return [engine.fetch_counts(ct) for ct in range(n_ct)], n_cols


def count_with(engine, n_ct, params, env):
    old = {k: os.environ.get(k) for k in ("LSG_NO_TM", "LSG_NO_INDEX", "LSG_TM_BYTES_PER_ENTRY", "LSG_LAYOUT")}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ["LSG_LAYOUT"] = "eager"
        os.environ.update(env)
        out = rows_of(engine, n_ct, params)
        return out, engine.layout_info()[0]
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def same(a, b):
    (ra, ca), (rb, cb) = a, b
    assert ca == cb
    for (k1, f1, c1), (k2, f2, c2) in zip(ra, rb):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(f1, f2); np.testing.assert_array_equal(c1, c2)


@pytest.mark.parametrize("n_ct", [1, 2])
def test_three_count_paths_write_the_same_rows(engine, n_ct):
    lens = [4000, 2000, 90]
    rec, refs, ct_of = make_case(11, 40000, lens, 400, n_ct=n_ct, hot_regions=[(0, 1000, 1100), (1, 500, 520)], hot_frac=0.8)
    load(engine, rec, lens, refs, ct_of, n_ct)
    p = CountParams.longsom_defaults()
    tm, path_tm = count_with(engine, n_ct, p, {})
    ix, path_ix = count_with(engine, n_ct, p, {"LSG_NO_TM": "1"})
    sc, path_sc = count_with(engine, n_ct, p, {"LSG_NO_INDEX": "1"})
    assert (path_tm, path_ix, path_sc) == (2, 1, 0)
    same(tm, ix); same(tm, sc)
    assert sum(len(k) for k, _, _ in tm[0]) > 0
    from oracle import loader                     # and all of them what the CPU oracle counts (the session's other tests reach the store form only)
    for ct in range(n_ct):
        ok, orf, oc, _ = loader.count(rec, lens, refs, ct_of, ct, p.min_bq, p.min_mq, p.min_dp, p.min_cc, p.flag_exclude, p.ignore_orphans)
        k, rf, c = sc[0][ct]
        np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(rf, orf); np.testing.assert_array_equal(c, oc)
    path, build_ms, store_bytes = engine.layout_info()
    assert build_ms > 0 and store_bytes > 0


def test_store_follows_filters_table_and_reads(engine):
    """new read filters rebuild the store, a new barcode table does not need to, new reads drop it: every count equals the scatter path's"""
    lens = [3000, 800]
    rec, refs, ct_of = make_case(12, 15000, lens, 120, hot_regions=[(0, 700, 760)], hot_frac=0.7)
    load(engine, rec, lens, refs, ct_of, 2)
    for p in (CountParams.longsom_defaults(), CountParams.longsom_defaults(min_mq=0, min_bq=0, min_dp=0, min_cc=0), CountParams.longsom_defaults(min_mq=30)):
        a, path = count_with(engine, 2, p, {})
        assert path == 2
        same(a, count_with(engine, 2, p, {"LSG_NO_INDEX": "1"})[0])
    # re-annotation: another table over the same reads (the store stays, the classes change)
    ct2 = ct_of.copy(); ct2[::3] = 1 - np.minimum(ct2[::3], 1); ct2[5] = 255
    p = CountParams.longsom_defaults()
    count_with(engine, 2, p, {})
    engine.set_barcodes(ct2, 2)
    built_before = engine.layout_info()[1]
    a, path = count_with(engine, 2, p, {})
    assert path == 2 and engine.layout_info()[1] == built_before
    same(a, count_with(engine, 2, p, {"LSG_NO_INDEX": "1"})[0])
    # other reads
    rec2, _, _ = make_case(13, 9000, lens, 120)
    engine.load_reads(rec2)
    a, path = count_with(engine, 2, p, {})
    assert path == 2
    same(a, count_with(engine, 2, p, {"LSG_NO_INDEX": "1"})[0])


def test_one_barcode_owning_a_tile_leaves_the_store(engine):
    """a single barcode's run of more entries than the planes' fields hold cannot be cut: such a load is counted on the tile index"""
    lens = [2500]
    rec, refs, ct_of = make_case(14, 30000, lens, 30, hot_regions=[(0, 700, 760)], hot_frac=0.97, cb_skew=0.9)
    ct_of[0] = 0
    load(engine, rec, lens, refs, ct_of, 2)
    p = CountParams.longsom_defaults()
    a, path = count_with(engine, 2, p, {})
    assert path == 1
    same(a, count_with(engine, 2, p, {"LSG_NO_INDEX": "1"})[0])


def test_region_counts_on_the_store_add_up(engine):
    lens = [6000, 1500]
    rec, refs, ct_of = make_case(15, 20000, lens, 150, hot_regions=[(0, 3000, 3100)], hot_frac=0.5)
    load(engine, rec, lens, refs, ct_of, 2)
    p = CountParams.longsom_defaults()
    whole, path = count_with(engine, 2, p, {})
    assert path == 2
    parts = []
    for lo, hi in (((0, 0), (0, 2944)), ((0, 2944), (1, 640)), ((1, 640), (2, 0))):
        engine.set_region(lo[0], lo[1], hi[0], hi[1])
        parts.append(count_with(engine, 2, p, {})[0])
    engine.set_region()
    for ct in range(2):
        k = np.concatenate([pr[0][ct][0] for pr in parts]); c = np.concatenate([pr[0][ct][2] for pr in parts])
        np.testing.assert_array_equal(k, whole[0][ct][0]); np.testing.assert_array_equal(c, whole[0][ct][2])


def test_a_load_too_large_for_the_store_is_counted_without_it(engine):
    """the store needs ~200 bytes of device memory per entry: when that is not free the count runs on the index (or the scatter)"""
    lens = [3000]
    rec, refs, ct_of = make_case(16, 8000, lens, 80)
    load(engine, rec, lens, refs, ct_of, 2)
    p = CountParams.longsom_defaults()
    a, path = count_with(engine, 2, p, {"LSG_TM_BYTES_PER_ENTRY": "2000000000"})
    assert path in (0, 1)
    b, path_b = count_with(engine, 2, p, {"LSG_NO_INDEX": "1"})
    assert path_b == 0
    same(a, b)
    engine.load_reads(rec)                      # the same reads again: nothing is remembered about the refusal
    c, path_c = count_with(engine, 2, p, {})
    assert path_c == 2
    same(a, c)


def test_default_policy_builds_the_store_for_a_load_that_keeps_being_counted(engine):
    """auto: a load counted one to three times never pays for the store; the fourth count under the same read filters builds it;
    lsg_prepare_counts builds it at once; never keeps every count on the scatter path"""
    lens = [3000, 500]
    rec, refs, ct_of = make_case(17, 9000, lens, 90)
    load(engine, rec, lens, refs, ct_of, 2)
    p = CountParams.longsom_defaults()
    first, path1 = count_with(engine, 2, p, {"LSG_LAYOUT": "auto"})
    assert path1 == 0 and engine.layout_info()[1] == 0.0
    for _ in range(2):
        again, path = count_with(engine, 2, p, {"LSG_LAYOUT": "auto"})
        assert path == 0 and engine.layout_info()[1] == 0.0
        same(first, again)
    second, path2 = count_with(engine, 2, p, {"LSG_LAYOUT": "auto"})
    third, path3 = count_with(engine, 2, p, {"LSG_LAYOUT": "auto"})
    assert (path2, path3) == (2, 2) and engine.layout_info()[1] > 0.0
    same(first, second); same(first, third)
    other = CountParams.longsom_defaults(min_mq=20)                      # other read filters, counted for the first time: no new store,
    o1, path4 = count_with(engine, 2, other, {"LSG_LAYOUT": "auto"})     # but the index (parameter-free) is there already
    assert path4 == 1
    same(o1, count_with(engine, 2, other, {"LSG_NO_INDEX": "1"})[0])
    engine.load_reads(rec)
    old = os.environ.pop("LSG_LAYOUT", None)
    try:
        os.environ["LSG_LAYOUT"] = "auto"
        engine.prepare_counts(p)
    finally:
        os.environ.pop("LSG_LAYOUT", None)
        if old is not None:
            os.environ["LSG_LAYOUT"] = old
    prepared, path5 = count_with(engine, 2, p, {"LSG_LAYOUT": "auto"})
    assert path5 == 2
    same(first, prepared)
    engine.load_reads(rec)
    for _ in range(3):
        never, path6 = count_with(engine, 2, p, {"LSG_LAYOUT": "never"})
        assert path6 == 0
    same(first, never)
