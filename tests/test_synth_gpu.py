"""Device-generated workload (synth.hip): counts on the GPU == events-level oracle on the copied-back arrays."""
import numpy as np
import pytest

from longsom_amd import synth
from longsom_amd._lib import CountParams

pytestmark = pytest.mark.gpu


def setup_model(engine, model):
    engine.set_contigs(model.contig_len)
    engine.synth_reference(model.seed)
    engine.set_barcodes(model.celltype_of, 2)
    engine.synth_reads(model)


def compare_with_oracle(engine, model, params=None):
    from oracle import loader
    params = params or CountParams.longsom_defaults()
    rec = engine.reads_to_host()
    refs = [engine.reference_to_host(t) for t in range(len(model.contig_len))]
    n_rows, n_cols = engine.pileup_count(params)
    tot = 0
    for ct in range(2):
        k, rf, c = engine.fetch_counts(ct)
        ok, orf, oc, ocols = loader.count(rec, model.contig_len, refs, model.celltype_of, ct, params.min_bq, params.min_mq,
                                          params.min_dp, params.min_cc, params.flag_exclude, params.ignore_orphans)
        tot += ocols
        np.testing.assert_array_equal(k, ok)
        np.testing.assert_array_equal(rf, orf)
        np.testing.assert_array_equal(c, oc)
    assert tot == n_cols
    return n_rows, n_cols, rec


def test_c1_device_generated(engine, kept_reads):
    model = synth.named("C1")
    setup_model(engine, model)
    rows, cols, rec = compare_with_oracle(engine, model)
    assert rec.n_reads == 50_000 and sum(rows) > 1000
    st = engine.count_stats()
    assert st.n_events_wave + st.n_events_deep == st.n_events_admitted
    assert st.n_events_admitted <= rec.n_events


def test_c2_small_with_chrM_deep(engine, kept_reads):
    """C2's genome and expression profile at 1/200 of the reads: chrM and the top genes go through the deep kernel."""
    model = synth.named("C2", n_reads=50_000, n_genes=2_000, n_cb=500)
    setup_model(engine, model)
    rows, cols, rec = compare_with_oracle(engine, model)
    assert engine.count_stats().n_deep_units > 0


def test_region_shards_partition_the_rows(engine):
    model = synth.named("C1", n_reads=20_000)
    setup_model(engine, model)
    engine.set_region()
    engine.pileup_count()
    full = [engine.fetch_counts(ct) for ct in range(2)]
    cut = (int(model.contig_len[0]) // 2) // 64 * 64
    parts = []
    for lo, hi in (((0, 0), (0, cut)), ((0, cut), (1, 0))):
        engine.set_region(lo[0], lo[1], hi[0], hi[1])
        engine.pileup_count()
        parts.append([engine.fetch_counts(ct) for ct in range(2)])
    engine.set_region()
    for ct in range(2):
        k = np.concatenate([parts[0][ct][0], parts[1][ct][0]])
        c = np.concatenate([parts[0][ct][2], parts[1][ct][2]])
        np.testing.assert_array_equal(k, full[ct][0])
        np.testing.assert_array_equal(c, full[ct][2])
        assert (parts[0][ct][0] & 0xffffffff).max() < cut <= (parts[1][ct][0] & 0xffffffff).min()


def test_device_generator_equals_host_model(engine, kept_reads):
    """synth.hip and the host evaluation of synth_model.h produce the same records, bit for bit"""
    from longsom_amd import hostio
    model = synth.named("C1", n_reads=3000, n_genes=60, n_cb=50)
    setup_model(engine, model)
    dev = engine.reads_to_host()
    host = hostio.synth_records(model)
    for name, _ in dev._SPEC:
        np.testing.assert_array_equal(getattr(dev, name), getattr(host, name), err_msg=name)


def test_tile_phased_generation_and_the_count_from_lines(engine, kept_reads):
    """LSG_LAYOUT_PHASED: the device generator == the host evaluation bit for bit (gaps are zeros); a load of such arrays that keeps no
    store bins them by 128-position windows and fetches every entry as one 256-byte block (lsg_get_layout_info path 6; by tiles, as one
    128-byte line, under LSG_NO_WINDOWS: path 5) and counts what the oracle counts"""
    from longsom_amd import hostio
    model = synth.named("C2", n_reads=40_000, n_genes=1_500, n_cb=400, layout=1)
    engine.set_contigs(model.contig_len); engine.synth_reference(model.seed); engine.set_barcodes(model.celltype_of, 2)
    cp = CountParams.longsom_defaults()
    saved = engine.load_settings()
    engine.set_count_at_load(cp); engine.set_store_policy(engine.STORE_SKIP_WHEN_COUNTED)
    engine.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)          # (the count's own read filter at load: keys alone through the sort)
    try:
        engine.synth_reads(model)
    finally:
        engine.restore_load_settings(saved)
    assert engine.layout_info()[0] == 6                      # (entries binned by 128-position windows: the generator's arrays are phased modulo 128)
    dev = engine.reads_to_host()
    host = hostio.synth_records(model)
    for name, _ in dev._SPEC:
        np.testing.assert_array_equal(getattr(dev, name), getattr(host, name), err_msg=name)
    assert (((dev.seg_ev_off - dev.seg_start) % 128) == 0).all()
    compare_with_oracle(engine, model, cp)
    assert engine.count_stats().n_deep_units > 0
    import os
    os.environ["LSG_NO_WINDOWS"] = "1"
    engine.set_count_at_load(cp); engine.set_store_policy(engine.STORE_SKIP_WHEN_COUNTED)
    try:
        engine.synth_reads(model)
    finally:
        engine.set_count_at_load(None); engine.set_store_policy(engine.STORE_KEEP); os.environ.pop("LSG_NO_WINDOWS")
    assert engine.layout_info()[0] == 5
    compare_with_oracle(engine, model, cp)
    # the same arrays through the store-keeping forms (k_tm_gather_count, then k_tm_gather + the walk): the keys carry lines there too
    engine.set_count_at_load(cp)
    try:
        engine.synth_reads(model)
    finally:
        engine.set_count_at_load(None)
    assert engine.layout_info()[0] == 3
    compare_with_oracle(engine, model, cp)
    engine.synth_reads(model)
    assert engine.layout_info()[0] == 2
    compare_with_oracle(engine, model, CountParams.longsom_defaults(min_bq=0))


def test_reads_are_not_kept_unless_asked(engine):
    """product mode: the tile store is the only resident copy of the events; the per-read and per-segment arrays come back"""
    from longsom_amd._lib import LsgError
    model = synth.named("C1", n_reads=2000, n_genes=40, n_cb=50)
    setup_model(engine, model)
    with pytest.raises(LsgError, match="not kept"):
        engine.reads_to_host()
