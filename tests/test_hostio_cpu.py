"""CPU: the synthetic BAM writer, the decoder and the host evaluation of the workload model agree; the BAM-level
oracle agrees with decoder + events-level oracle on a model BAM (indels, introns, clips, all flag classes)."""
import numpy as np

from longsom_amd import hostio, synth
from oracle import loader


def small_model():
    return synth.named("C1", n_reads=1500, n_genes=40, n_cb=60)


def test_bam_roundtrip_equals_model_records(tmp_path):
    m = small_model()
    bam = str(tmp_path / "s.bam")
    hostio.synth_bam(m, bam, str(tmp_path / "s.fa"))
    bcs = hostio.synth_barcodes(m)
    assert len(set(bcs)) == m.n_cb
    dec = hostio.decode_bam(bam, bcs, min_mapq=60)
    rec = hostio.synth_records(m)
    keep = rec.read_cb >= 0                       # the decoder drops reads without a listed CB
    want = rec.subset(keep)
    got = dec.records
    assert got.n_reads == want.n_reads and got.n_events == want.n_events
    # BAM order is (tid, pos); compare per read through a canonical sort
    def canon(r):
        first = np.zeros(r.n_reads, np.int64); first[r.seg_read[::-1]] = r.seg_ev_off[::-1]
        nev = np.bincount(r.seg_read, weights=r.seg_len, minlength=r.n_reads).astype(np.int64)
        sig = [(int(r.read_tid[i]), int(r.read_pos[i]), int(r.read_flag[i]), int(r.read_mapq[i]), int(r.read_cb[i]),
                r.events[first[i]:first[i] + nev[i]].tobytes()) for i in range(r.n_reads)]
        return sorted(sig)
    assert canon(got) == canon(want)
    assert dec.report["Total_reads"] == m.n_reads
    assert dec.report["CB_not_found"] == int((rec.read_cb < 0).sum()) - dec.report["CB_not_matched"]


def test_tile_phased_layout_of_the_model_records():
    """LSG_LAYOUT_PHASED (include/longsom_hip.h): the same reads, segments and events; every segment at an offset congruent to its reference
    start modulo 64, every read's region a multiple of 64 events, zeros between the segments; the events-level oracle counts them the same"""
    from tests.util import assert_same_records, phased_records
    m = synth.named("C1", n_reads=1500, n_genes=40, n_cb=60)
    compact = hostio.synth_records(m)
    m.layout = 1
    phased = hostio.synth_records(m)
    assert_same_records(phased, compact, phased_a=True)
    assert phased.n_events > compact.n_events
    again = phased_records(compact)                    # (the test helper that lays any compact arrays out the same way)
    np.testing.assert_array_equal(again.seg_ev_off, phased.seg_ev_off); np.testing.assert_array_equal(again.events, phased.events)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    for ct in range(2):
        a, b = loader.count(compact, m.contig_len, refs, m.celltype_of, ct), loader.count(phased, m.contig_len, refs, m.celltype_of, ct)
        for x, y in zip(a[:3], b[:3]):
            np.testing.assert_array_equal(x, y)
        assert a[3] == b[3]


def test_plp_oracle_matches_decoder_path_on_model_bam(tmp_path):
    m = small_model()
    bam = str(tmp_path / "s.bam")
    hostio.synth_bam(m, bam, str(tmp_path / "s.fa"), barcode_suffix="-1")
    bcs = hostio.synth_barcodes(m)
    dec = hostio.decode_bam(bam, bcs, min_mapq=60)
    from longsom_amd import tsvio
    names, refs = tsvio.read_fasta(str(tmp_path / "s.fa"))
    assert names == m.contig_names
    for ct in (0, 1):
        for params in (dict(min_bq=20, min_mq=60, min_dp=5, min_cc=5), dict(min_bq=10, min_mq=30, min_dp=2, min_cc=1)):
            k, r, c = loader.plp_count(bam, bcs, m.celltype_of, ct, m.contig_len, refs, **params)
            k2, r2, c2, _ = loader.count(dec.records, m.contig_len, refs, m.celltype_of, ct, params["min_bq"], params["min_mq"],
                                         params["min_dp"], params["min_cc"])
            assert len(k) > 0
            np.testing.assert_array_equal(k, k2); np.testing.assert_array_equal(r, r2); np.testing.assert_array_equal(c, c2)
