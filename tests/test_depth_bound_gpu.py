"""GPU: the load-time bound that stands in for the reference's pileup depth cap (bam.pileup(..., max_depth = 200000),
BaseCellCounter.py:191).  The bound must never under-count the reads live at one position, and the host mirror must refuse a
sample above the cap unless told otherwise."""
import numpy as np
import pytest

from longsom_amd import hostio, pipeline, synth
from longsom_amd.synth_simple import random_records, random_reference

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
            st = int(rec.seg_start[segs[0]]); en = int(rec.seg_start[segs[-1]] + rec.seg_len[segs[-1]] - 1)
            diff[min(st >> 6, nt - 1)] += 1; diff[min((en >> 6) + 1, nt)] -= 1
            pdiff[st] += 1; pdiff[en + 1] -= 1
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
    # the bound follows the barcode table: everything in one cell type = the union of the reads with a barcode
    engine.set_barcodes(np.zeros(40, np.uint8), 1)
    assert engine.max_live_reads() == tile_bound(rec, lens, has_cb)[0] >= got


def test_pipeline_refuses_a_sample_above_the_cap(engine, tmp_path, monkeypatch, capsys):
    m = synth.named("C1", n_reads=4000, n_genes=4, n_cb=20, snp_mod=120)
    bam, fa, bct = str(tmp_path / "S1.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa)
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    res = pipeline.load_sample(bam, bct, fa, engine, 60)          # far below 200000: loads
    live = engine.max_live_reads()
    assert 0 < live <= 4000
    monkeypatch.setattr(pipeline, "PILEUP_MAX_DEPTH", live - 1)
    monkeypatch.delenv("LONGSOM_ALLOW_DEPTH_OVERFLOW", raising=False)
    with pytest.raises(pipeline.DepthCapExceeded):
        pipeline.load_sample(bam, bct, fa, engine, 60)
    monkeypatch.setenv("LONGSOM_ALLOW_DEPTH_OVERFLOW", "1")
    res = pipeline.load_sample(bam, bct, fa, engine, 60)
    assert "max_depth" in capsys.readouterr().err
    assert res.engine is engine
