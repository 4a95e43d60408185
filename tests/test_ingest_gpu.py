"""GPU: device-side BAM ingest (lsg_load_bam: BGZF inflate, record chain, CB lookup, SplitBam's counters, CIGAR walk on the GPU)
against the host decoder (hostio.decode_bam), which is pinned to the reference-run goldens: same report, same per-barcode tallies,
same read-record arrays, same count tables."""
import os

import numpy as np
import pytest

from longsom_amd import hostio, synth
from longsom_amd._lib import CountParams, LsgError
from tests.util import assert_same_records

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def both_ways(engine, bam, barcodes, celltype_of, refs, min_mapq=60, compact=False):
    """loads `bam` on the device and through the host decoder; returns (device info, cb_pass, cb_low, DecodedBam)"""
    names, lens, first = hostio.bam_header(bam)
    dec = hostio.decode_bam(bam, barcodes, min_mapq=min_mapq)
    assert names == dec.contig_names and list(lens) == list(dec.contig_len)
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(celltype_of, 2)
    engine.set_region()
    engine.set_keep_reads(True)
    try:
        info, cb_pass, cb_low = engine.load_bam(bam, barcodes, min_mapq=min_mapq, first_record_offset=first)
        dev = engine.reads_to_host()
    finally:
        engine.set_keep_reads(False)
    rep = {"Total_reads": info["total_reads"], "Pass_reads": info["pass_reads"], "CB_not_found": info["cb_not_found"], "CB_not_matched": info["cb_not_matched"]}
    if info["mapq_filtered"]:
        rep["MAPQ"] = info["mapq_filtered"]
    assert rep == dec.report
    np.testing.assert_array_equal(cb_pass, dec.cb_pass); np.testing.assert_array_equal(cb_low, dec.cb_low)
    if compact:                                                  # (LSG_INGEST_COMPACT: the host decoder's layout, array for array)
        for name, _ in dev._SPEC:
            np.testing.assert_array_equal(getattr(dev, name), getattr(dec.records, name), err_msg=name)
    else:                                                        # the device decoder lays the events out tile-phased (LSG_LAYOUT_PHASED)
        assert_same_records(dev, dec.records, phased_a=True)
    rows_dev = (engine.pileup_count(), [engine.fetch_counts(ct) for ct in range(2)])
    engine.load_reads(dec.records)
    rows_host = (engine.pileup_count(), [engine.fetch_counts(ct) for ct in range(2)])
    assert rows_dev[0] == rows_host[0]
    for (k1, r1, c1), (k2, r2, c2) in zip(rows_dev[1], rows_host[1]):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(c1, c2)
    return info, dec


@pytest.mark.parametrize("tag", ["rand", "randsfx"])
def test_reference_pinned_sample(engine, tag):
    """the multi-contig sample of the reference-run goldens (every CIGAR operation, all flags, CB missing / unknown / suffixed); written by
    the Python BAM writer, whose records straddle its BGZF blocks: the record chain needs its fix-up rounds"""
    from longsom_amd import tsvio
    bc = hostio.read_barcodes(os.path.join(G, "pileup.%s.barcodes.tsv" % tag))
    names, seqs = tsvio.read_fasta(os.path.join(G, "pileup.rand.fa"))
    refs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), dtype=np.uint8) for s in seqs]
    info, dec = both_ways(engine, os.path.join(G, "pileup.%s.bam" % tag), bc.barcodes, bc.celltype_of, refs)
    assert info["n_records"] == 1418 and info["chain_rounds"] >= 1


def test_synthetic_bam_like_htslib(engine, tmp_path):
    """a BAM written the way htslib writes (no record straddles a block): one round settles the chain; C1's model, 20 k reads"""
    m = synth.named("C1", n_reads=20_000)
    bam = str(tmp_path / "c1.bam")
    hostio.synth_bam(m, bam)
    barcodes = hostio.synth_barcodes(m)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    info, dec = both_ways(engine, bam, barcodes, m.celltype_of, refs)
    assert info["chain_rounds"] <= 2 and info["n_blocks"] > 100
    assert info["n_records"] == 20_000


def test_legacy_del_merge_and_low_mapq(engine, tmp_path):
    m = synth.named("C1", n_reads=3_000)
    bam = str(tmp_path / "c1.bam")
    hostio.synth_bam(m, bam)
    barcodes = hostio.synth_barcodes(m)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    old = hostio.set_legacy_del_merge(True)
    try:
        both_ways(engine, bam, barcodes, m.celltype_of, refs, min_mapq=30)
        os.environ["LSG_INGEST_COMPACT"] = "1"                   # the device decoder's compact layout: the host decoder's arrays exactly
        both_ways(engine, bam, barcodes, m.celltype_of, refs, min_mapq=30, compact=True)
    finally:
        os.environ.pop("LSG_INGEST_COMPACT", None)
        hostio.set_legacy_del_merge(old)


def test_damaged_files_are_errors(engine, tmp_path):
    raw = open(os.path.join(G, "pileup.rand.bam"), "rb").read()
    bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
    names, lens, first = hostio.bam_header(os.path.join(G, "pileup.rand.bam"))
    engine.set_contigs(lens); engine.set_barcodes(bc.celltype_of, 2)
    rng = np.random.default_rng(3)
    cases = {"cut.bam": raw[: len(raw) // 2], "cut2.bam": raw[:-40]}
    for k in range(6):                                  # bit flips inside the compressed payload of some block
        b = bytearray(raw); at = int(rng.integers(200, len(raw) - 60)); b[at] ^= 1 << int(rng.integers(0, 8)); cases["flip%d.bam" % k] = bytes(b)
    n_err = 0
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        try:
            engine.load_bam(str(p), bc.barcodes, first_record_offset=first)
        except LsgError:
            n_err += 1
    assert n_err >= 2                                   # the truncated ones always; a flipped bit may land in a CRC or in padding
    # the handle is usable afterwards
    engine.load_bam(os.path.join(G, "pileup.rand.bam"), bc.barcodes, first_record_offset=first)
    assert engine.reads_shape()[0] == 1333


@pytest.mark.parametrize("source", ["rand", "synth"])
@pytest.mark.parametrize("world", [2, 5])
def test_slices_of_an_indexed_bam_add_up_to_the_whole(engine, tmp_path, source, world):
    """lsg_load_bam_range over the slices regions.BaiPlan cuts from the .bai's linear index: every region's rows are the whole file's
    rows inside the region, SplitBam's counters and the per-barcode tallies add up to the whole file's — on the reference-pinned sample
    (records straddle its blocks: a slice starts INSIDE a block, its last record is cut) and on an htslib-like 20 k-read BAM"""
    import shutil
    from longsom_amd import regions, tsvio
    bam = str(tmp_path / "a.bam")
    if source == "rand":
        shutil.copy(os.path.join(G, "pileup.rand.bam"), bam)
        bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
        barcodes, celltype_of = bc.barcodes, bc.celltype_of
        _, seqs = tsvio.read_fasta(os.path.join(G, "pileup.rand.fa"))
        refs = [np.frombuffer(bytes(x), dtype=np.uint8) for x in seqs]
    else:
        m = synth.named("C1", n_reads=20000, n_genes=50, n_cb=80, snp_mod=150)
        hostio.synth_bam(m, bam, str(tmp_path / "ref.fa"))
        barcodes, celltype_of = hostio.synth_barcodes(m), m.celltype_of
        _, seqs = tsvio.read_fasta(str(tmp_path / "ref.fa"))
        refs = [np.frombuffer(bytes(x), dtype=np.uint8) for x in seqs]
    names, lens, first = hostio.bam_header(bam)
    engine.set_contigs(lens)
    for t, r in enumerate(refs):
        engine.load_reference(t, r)
    engine.set_barcodes(celltype_of, 2)
    engine.set_region()
    info_w, pass_w, low_w = engine.load_bam(bam, barcodes, min_mapq=60, first_record_offset=first)
    engine.pileup_count()
    whole = [engine.fetch_counts(ct) for ct in range(2)]
    plan = regions.BaiPlan(hostio.read_bai(hostio.build_bai(bam)), len(names), world, os.path.getsize(bam))
    keys = ("total_reads", "pass_reads", "cb_not_found", "cb_not_matched", "mapq_filtered")
    tot = dict.fromkeys(keys, 0); cb_pass = np.zeros_like(pass_w); cb_low = np.zeros_like(low_w)
    parts = [[], []]
    for r in range(world):
        lo, hi = plan.bounds[r], plan.bounds[r + 1]
        got = regions.ingest_slice(engine, bam, plan, lo, hi, barcodes, 60)
        if got is None:
            continue
        info, p, l = got
        for k in keys:
            tot[k] += info[k]
        cb_pass += p; cb_low += l
        engine.set_region(lo[0], lo[1], hi[0], hi[1])
        engine.pileup_count()
        for ct in range(2):
            parts[ct].append(engine.fetch_counts(ct))
    engine.set_region()
    assert tot == {k: info_w[k] for k in keys}
    np.testing.assert_array_equal(cb_pass, pass_w); np.testing.assert_array_equal(cb_low, low_w)
    for ct in range(2):
        for j in range(3):
            np.testing.assert_array_equal(np.concatenate([x[j] for x in parts[ct]]), whole[ct][j])


def test_the_whole_device_ingest_path_at_scale(engine, tmp_path):
    """C2 at 1.5 M reads written as a BAM (1.5 GB, 24 k BGZF blocks) -> pipeline.run_snv, whose default is the device ingest (inflate, record
    chain, decode, tile-phased events, the count inside the load, no store): the count rows == the digests the CPU oracle wrote for this
    workload (tests/golden/rows_hash_oracle_c2_1500000.json: tools/oracle_hashes.py over hostio.synth_records), and the read-record arrays
    the device decoder leaves == the host decoder's for the same file, read for read and event for event"""
    import json
    from longsom_amd import pipeline
    m = synth.named("C2", n_reads=1_500_000)
    bam, fa, bct = str(tmp_path / "S.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "bc.tsv")
    hostio.synth_bam(m, bam, fa)
    barcodes = hostio.synth_barcodes(m)
    hostio.write_barcodes_tsv(bct, barcodes, m.celltype_of, ["Cancer", "Non-Cancer"])
    engine.unload_reads()                                        # (the run below brings its own handle, as the rule's script does: room for both)
    out = pipeline.run_snv(bam, bct, fa, str(tmp_path / "out"), "S", params=pipeline.SnvParams(row_digests=True))
    assert any(k.startswith("ingest_") for k in out.timings), "the run took the host decoder"
    want = json.load(open(os.path.join(G, "rows_hash_oracle_c2_1500000.json")))
    got = out.row_digests
    assert got["rows"] == want["rows"] and got["columns"] == want["columns"]
    for ct in range(2):
        assert got["ct%d" % ct] == want["ct%d" % ct], "count rows of cell type %d differ from the CPU oracle's" % ct
    # the arrays themselves, device decoder against host decoder
    names, lens, first = hostio.bam_header(bam)
    engine.set_contigs(lens); engine.set_barcodes(m.celltype_of, 2); engine.set_region()
    engine.set_keep_reads(True)
    try:
        info, _, _ = engine.load_bam(bam, barcodes, min_mapq=60, first_record_offset=first)
        dev = engine.reads_to_host()
    finally:
        engine.set_keep_reads(False)
    dec = hostio.decode_bam(bam, barcodes, min_mapq=60)
    assert info["n_records"] == 1_500_000 and dev.n_reads == dec.records.n_reads
    assert_same_records(dev, dec.records, phased_a=True)
    engine.unload_reads()


def _bgzf_blocks(raw):
    out, off = [], 0
    while off < len(raw):
        xlen = int.from_bytes(raw[off + 10:off + 12], "little")
        bsize = None
        q = off + 12
        while q < off + 12 + xlen:
            slen = int.from_bytes(raw[q + 2:q + 4], "little")
            if raw[q:q + 2] == b"BC" and slen == 2:
                bsize = int.from_bytes(raw[q + 4:q + 6], "little") + 1
            q += 4 + slen
        out.append((off, bsize, xlen))
        off += bsize
    return out


def test_a_block_whose_payload_inflates_to_isize_with_a_wrong_crc_is_refused(engine, tmp_path):
    """htslib checks every BGZF block's CRC32 (pysam refuses such a file: SplitBamCellTypes.py:51,65): a block whose payload was changed -
    one quality value - and deflated again, under the OLD trailer, still inflates to ISIZE bytes; both decoders must say no, and say why"""
    import zlib
    src = os.path.join(G, "pileup.rand.bam")
    raw = open(src, "rb").read()
    bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
    names, lens, first = hostio.bam_header(src)
    blocks = _bgzf_blocks(raw)
    off, bsize, xlen = max(blocks[:-1], key=lambda b: b[1])          # the largest block (not the EOF marker)
    payload = raw[off + 12 + xlen:off + bsize - 8]
    data = bytearray(zlib.decompress(payload, -15))
    assert len(data) == int.from_bytes(raw[off + bsize - 4:off + bsize], "little") and zlib.crc32(bytes(data)) == int.from_bytes(raw[off + bsize - 8:off + bsize - 4], "little")
    data[len(data) // 2] ^= 0x04
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    new_payload = co.compress(bytes(data)) + co.flush()
    new_bsize = 12 + xlen + len(new_payload) + 8
    head = bytearray(raw[off:off + 12 + xlen])
    q = 12
    while q < 12 + xlen:
        slen = int.from_bytes(head[q + 2:q + 4], "little")
        if head[q:q + 2] == b"BC":
            head[q + 4:q + 6] = (new_bsize - 1).to_bytes(2, "little")
        q += 4 + slen
    bad = raw[:off] + bytes(head) + new_payload + raw[off + bsize - 8:off + bsize] + raw[off + bsize:]
    p = tmp_path / "badcrc.bam"
    p.write_bytes(bad)
    engine.set_contigs(lens); engine.set_barcodes(bc.celltype_of, 2)
    with pytest.raises(LsgError, match="CRC32"):
        engine.load_bam(str(p), bc.barcodes, first_record_offset=first)
    with pytest.raises(Exception, match="CRC32"):
        hostio.decode_bam(str(p), bc.barcodes, min_mapq=60)
    # ... and the untouched file still loads, on the same handle (every one of its blocks' CRCs agrees)
    engine.load_bam(src, bc.barcodes, first_record_offset=first)
    assert engine.reads_shape()[0] == 1333
