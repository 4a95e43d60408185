"""GPU: the reference's pileup depth cap (bam.pileup(..., max_depth = 200000), BaseCellCounter.py:191).
  * lsg_max_live_reads: the bound that tells the count path whether the cap can fire at all must never under-count the reads
    buffered at one position (a read stays buffered through the column after its last one);
  * lsg_pileup_count with max_depth below the depth of the sample drops exactly the reads htslib's rule drops: rows equal the
    BAM-level oracle's (oracle/plp_oracle.c, whose rule is pinned by the reference-written tests/golden/pileup.cap.* tables);
  * the genotyping path does not model the cap: its guard refuses a sample above it unless told otherwise."""
import numpy as np
import pytest

from longsom_amd import hostio, pipeline, synth
from tests.support.synth_simple import random_records, random_reference

pytestmark = pytest.mark.gpu


def tile_bound(rec, contig_lens, keep):
    """numpy restatement: reads selected by keep[read], span = first segment start .. last segment end, counted per 64-position tile."""
    best = 0
    exact = 0
    sr = rec.seg_read.astype(np.int64)
    for tid, L in enumerate(contig_lens):
        nt = (int(L) + 63) // 64
        diff = np.zeros(nt + 1, np.int64)
        pdiff = np.zeros(int(L) + 2, np.int64)
        for r in np.flatnonzero((rec.read_tid == tid) & keep):
            segs = np.flatnonzero(sr == r)
            if len(segs) == 0:
                continue
            st = int(rec.seg_start[segs[0]]); en = int(rec.seg_start[segs[-1]] + rec.seg_len[segs[-1]])      # buffered through column `en`
            diff[min(st >> 6, nt - 1)] += 1; diff[min((en >> 6) + 1, nt)] -= 1
            pdiff[st] += 1; pdiff[min(en, int(L)) + 1] -= 1
        best = max(best, int(np.cumsum(diff).max()))
        exact = max(exact, int(np.cumsum(pdiff).max()))
    return best, exact


@pytest.mark.parametrize("seed,lens,hot", [(11, [5000, 1200, 70], None), (12, [4000, 2000], [(0, 1000, 1100), (1, 500, 520)])])
def test_bound_equals_the_tile_restatement_and_covers_the_true_depth(engine, seed, lens, hot):
    kw = dict(hot_regions=hot, hot_frac=0.8) if hot else {}
    rec = random_records(seed, 1500, lens, 40, **kw)
    rng = np.random.default_rng(seed)
    engine.set_contigs(lens)
    for t, L in enumerate(lens):
        engine.load_reference(t, random_reference(rng, int(L)))
    ct_of = rng.integers(0, 3, 40).astype(np.uint8)
    ct_of[ct_of == 2] = 255                                      # a third of the barcodes unused
    engine.set_barcodes(ct_of, 2)
    engine.load_reads(rec)
    has_cb = rec.read_cb >= 0
    ct_read = np.where(has_cb, ct_of[np.maximum(rec.read_cb, 0)], 255)
    per_ct = [tile_bound(rec, lens, ct_read == ct) for ct in range(2)]
    got = engine.max_live_reads()
    assert got == max(b for b, _ in per_ct)
    assert got >= max(e for _, e in per_ct) > 0
    # the bound over ALL reads with a barcode (what the genotyping pileup of the unsplit BAM holds): table-independent, falls out of the load
    assert engine.max_live_reads_all() == tile_bound(rec, lens, has_cb)[0] >= got
    # the bound follows the barcode table: everything in one cell type = the union of the reads with a barcode
    engine.set_barcodes(np.zeros(40, np.uint8), 1)
    assert engine.max_live_reads() == tile_bound(rec, lens, has_cb)[0] >= got


@pytest.fixture(scope="module")
def deep_sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("deep")
    m = synth.named("C1", n_reads=6000, n_genes=5, n_cb=40, snp_mod=120)
    bam, fa, bct = str(d / "S1.bam"), str(d / "ref.fa"), str(d / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa)
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    return m, bam, fa, bct


@pytest.mark.parametrize("max_depth", [1, 37, 150, 400, 200000])
def test_count_with_a_depth_cap_equals_the_bam_level_oracle(engine, deep_sample, max_depth):
    from longsom_amd import tsvio
    from longsom_amd._lib import CountParams
    from oracle import loader
    m, bam, fa, bct = deep_sample
    res = pipeline.load_sample(bam, bct, fa, engine, 60)
    names, seqs = tsvio.read_fasta(fa)
    live = engine.max_live_reads()
    assert live > 400                                             # the caps above bite
    rows, _ = engine.pileup_count(CountParams.longsom_defaults(max_depth=max_depth))
    uncapped, _ = engine.pileup_count(CountParams.longsom_defaults(max_depth=0))
    dp_uncapped = sum(int(engine.fetch_counts(ct)[2][:, 0].sum()) for ct in range(2))
    if max_depth >= live:
        assert rows == uncapped
    engine.pileup_count(CountParams.longsom_defaults(max_depth=max_depth))
    dp = 0
    for ct in range(2):
        k, r, c = engine.fetch_counts(ct)
        dp += int(c[:, 0].sum())
        ok, orf, oc = loader.plp_count(bam, res.table.barcodes, res.table.celltype_of, ct, [len(s) for s in seqs], seqs, max_depth=max_depth)
        assert np.array_equal(k, ok) and np.array_equal(r, orf) and np.array_equal(c, oc), "cell type %d, max_depth %d" % (ct, max_depth)
    if max_depth <= 150:
        assert dp < dp_uncapped                                   # the cap really dropped reads
    else:
        assert dp <= dp_uncapped


def test_the_position_level_bound_is_where_the_cap_stops_biting(engine, deep_sample):
    """lsg_max_live_reads_exact = B: with max_depth >= B nothing is dropped (the host replay is skipped), and the BAM-level oracle agrees
    on both sides of B - below it the replay runs and drops what htslib drops"""
    from longsom_amd import tsvio
    from longsom_amd._lib import CountParams
    from oracle import loader
    m, bam, fa, bct = deep_sample
    res = pipeline.load_sample(bam, bct, fa, engine, 60)
    names, seqs = tsvio.read_fasta(fa)
    B = engine.max_live_reads_exact()
    assert 8 < B <= engine.max_live_reads() <= engine.max_live_reads_all()      # the tiles over-count
    uncapped, _ = engine.pileup_count(CountParams.longsom_defaults(max_depth=0))
    for max_depth in (B + 1, B, B - 1, B - 2, B // 2):
        rows, _ = engine.pileup_count(CountParams.longsom_defaults(max_depth=max_depth))
        if max_depth >= B:
            assert rows == uncapped
        for ct in range(2):
            k, r, c = engine.fetch_counts(ct)
            ok, orf, oc = loader.plp_count(bam, res.table.barcodes, res.table.celltype_of, ct, [len(s) for s in seqs], seqs, max_depth=max_depth)
            assert np.array_equal(k, ok) and np.array_equal(r, orf) and np.array_equal(c, oc), "cell type %d, max_depth %d (B = %d)" % (ct, max_depth, B)


@pytest.mark.parametrize("window,max_depth", [(150, 37), (64, 5), (1000, 150)])
def test_the_cap_is_replayed_per_pileup_window(engine, deep_sample, window, max_depth):
    """every window [1 + k W, 1 + (k + 1) W) is a pileup of its own (BaseCellCounter.py:185-191): narrow windows put an edge into most tiles
    of the deep genes - entries cut at the edges, reads dropped in one window and counted in the next - and the counts equal the
    BAM-level oracle's, which sweeps window by window"""
    from longsom_amd import tsvio
    from longsom_amd._lib import CountParams
    from oracle import loader
    m, bam, fa, bct = deep_sample
    engine.set_pileup_window(window)
    try:
        res = pipeline.load_sample(bam, bct, fa, engine, 60)
        names, seqs = tsvio.read_fasta(fa)
        engine.pileup_count(CountParams.longsom_defaults(max_depth=max_depth))
        differs = False
        for ct in range(2):
            k, r, c = engine.fetch_counts(ct)
            ok, orf, oc = loader.plp_count(bam, res.table.barcodes, res.table.celltype_of, ct, [len(s) for s in seqs], seqs, max_depth=max_depth, window=window)
            assert np.array_equal(k, ok) and np.array_equal(r, orf) and np.array_equal(c, oc), "cell type %d" % ct
            k1, _, c1 = loader.plp_count(bam, res.table.barcodes, res.table.celltype_of, ct, [len(s) for s in seqs], seqs, max_depth=max_depth, window=0)
            differs |= not (np.array_equal(k, k1) and np.array_equal(c, c1))
        assert differs                                            # one stream per contig would have counted something else
        # without a cap the windows change nothing: the rows of the entries cut at the edges equal the events-level oracle's
        engine.pileup_count(CountParams.longsom_defaults(max_depth=0))
        for ct in range(2):
            k, r, c = engine.fetch_counts(ct)
            ok, orf, oc = loader.plp_count(bam, res.table.barcodes, res.table.celltype_of, ct, [len(s) for s in seqs], seqs, max_depth=0)
            assert np.array_equal(k, ok) and np.array_equal(c, oc)
    finally:
        engine.set_pileup_window(50000)
