"""CPU: the region machinery of the sharded / windowed SNV chain (longsom_amd/regions.py, pipeline._windows), checked with the CPU
oracle standing in for the device: counting region by region — with the boundary-crossing reads loaded on both sides — gives the rows
of one count over everything; pieces are concatenated in the reference's (chrom string, start) order; the byte all-gather works over
a 2-rank gloo group."""
import os
import socket

import numpy as np
import pytest

from longsom_amd import hostio, pipeline, regions, tsvio
from oracle import loader

G = os.path.join(os.path.dirname(__file__), "golden")
BAM = os.path.join(G, "pileup.rand.bam")


def inputs():
    bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
    names, seqs = tsvio.read_fasta(os.path.join(G, "pileup.rand.fa"))
    return bc, names, seqs, [len(s) for s in seqs]


def rows_in(rec, lens, refs, bc, ct, lo, hi):
    k, r, c, _ = loader.count(rec, lens, refs, bc.celltype_of, ct)
    m = (k >= pipeline._key(lo)) & (k < pipeline._key(hi))
    return k[m], r[m], c[m]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_event_balanced_regions_reproduce_the_whole_count(world):
    bc, names, refs, lens = inputs()
    dec = hostio.decode_bam(BAM, bc.barcodes)
    b = regions.balanced_boundaries(dec.records, len(names), world)
    assert b[0] == (0, 0) and b[-1] == (len(names), 0) and len(b) == world + 1
    assert all(pipeline._key(x) <= pipeline._key(y) for x, y in zip(b, b[1:])) and all(p % 64 == 0 for _, p in b)
    ev = regions.read_events(dec.records)
    for ct in range(2):
        whole = loader.count(dec.records, lens, refs, bc.celltype_of, ct)
        parts = []
        for r in range(world):
            mine = dec.records.subset(regions.reads_overlapping(dec.records, b[r], b[r + 1]))
            parts.append(rows_in(mine, lens, refs, bc, ct, b[r], b[r + 1]))
        for i in range(3):
            assert np.array_equal(np.concatenate([p[i] for p in parts]), whole[i])
    # balance: no rank holds more than twice its share of the events (the sample has a 60 kb contig and three small ones)
    if world <= 3:
        start = (dec.records.read_tid.astype(np.int64) << 32) | dec.records.read_pos
        share = [ev[(start >= pipeline._key(b[r])) & (start < pipeline._key(b[r + 1]))].sum() for r in range(world)]
        assert max(share) < 2.0 * ev.sum() / world


@pytest.mark.parametrize("batch", [2000, 70000, 140000])
def test_windows_with_carried_reads_reproduce_the_whole_count(batch):
    bc, names, refs, lens = inputs()
    whole_dec = hostio.decode_bam(BAM, bc.barcodes)
    wins = list(pipeline._windows(hostio.stream_bam(BAM, bc.barcodes, batch_bytes=batch), len(names)))
    assert len(wins) > 1 and wins[0][0] == (0, 0) and wins[-1][1] == (len(names), 0)
    assert all(a[1] == b[0] for a, b in zip(wins, wins[1:]))
    report = {}
    for _, _, _, dec in wins:
        for k, v in dec.report.items():
            report[k] = report.get(k, 0) + v
    assert report == whole_dec.report
    for ct in range(2):
        whole = loader.count(whole_dec.records, lens, refs, bc.celltype_of, ct)
        parts = [rows_in(rec, lens, refs, bc, ct, lo, hi) for lo, hi, rec, _ in wins]
        for i in range(3):
            assert np.array_equal(np.concatenate([p[i] for p in parts]), whole[i])


def test_prefetch_hands_over_items_and_errors():
    assert list(pipeline._prefetch(iter(range(5)))) == [0, 1, 2, 3, 4]

    def boom():
        yield 1
        raise KeyError("x")
    g = pipeline._prefetch(boom())
    assert next(g) == 1
    with pytest.raises(KeyError):
        next(g)


def test_pieces_are_concatenated_in_chrom_string_then_start_order(tmp_path):
    for chrom, start in (("chr2", 5), ("chr10", 700), ("chr1", 50001), ("chr1", 2), ("chrM", 1), ("chr10", 64)):
        open(regions.piece_path(str(tmp_path), chrom, start, "counts.Non-Cancer"), "w").write("%s\t%d\n" % (chrom, start))
    open(regions.piece_path(str(tmp_path), "chr1", 1, "merged"), "w").write("other table\n")
    out = tmp_path / "o.tsv"
    assert regions.concatenate_pieces(str(tmp_path), "counts.Non-Cancer", "#h\n", str(out)) == 6
    assert out.read_text() == "#h\nchr1\t2\nchr1\t50001\nchr10\t64\nchr10\t700\nchr2\t5\nchrM\t1\n"        # BaseCellCounter.py:64-70


def _rank_main(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LSG_DIST_BACKEND="gloo")
    comm = regions.Comm.from_env()
    kept = {("chr%d" % (rank + 1), 100 * rank + 1): "row of rank %d\n" % rank, ("chr1", 5000 + rank): "late row %d\n" % rank} if rank else {}
    got = comm.allgather_bytes(regions.pack_rows(kept))
    total = comm.allreduce_sum(np.asarray([rank + 1, 10 * (rank + 1), 0], np.int64))       # (SplitBam's counters of a sliced ingest are summed this way)
    assert total.tolist() == [world * (world + 1) // 2, 10 * world * (world + 1) // 2, 0]
    comm.barrier()
    q.put((rank, regions.unpack_rows(got)))
    comm.close()


def test_allgather_bytes_over_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = "late row 1\nlate row 2\nrow of rank 1\nrow of rank 2\n"       # (chr1, 5001), (chr1, 5002), (chr2, 101), (chr3, 201)
    assert res == {0: want, 1: want, 2: want}


def _rank_agree(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LSG_DIST_BACKEND="gloo")
    comm = regions.Comm.from_env()
    comm.agree(None, "a part nobody fails")                          # a vote everybody passes
    try:
        comm.agree(ValueError("index of another BAM") if rank == 1 else None, "the ingest of a rank's slice")
        q.put((rank, "went on"))
    except ValueError as e:
        q.put((rank, "own error: %s" % e))
    except RuntimeError as e:
        q.put((rank, "told: %s" % e))
    comm.barrier()                                                    # every rank is still in step: nobody waits for one that left
    comm.close()


def test_a_rank_that_fails_between_collectives_takes_every_rank_out_together():
    """regions.Comm.agree: the vote before the ranks go on (a failed slice ingest, a failed region) - no rank blocks in the next collective"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_rank_agree, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[1] == "own error: index of another BAM"
    assert res[0] == res[2] == "told: the ingest of a rank's slice failed on 1 of the 3 ranks (their own errors say why)"


def test_bai_reader_skips_the_binning_index_and_the_builder_round_trips(tmp_path):
    """read_bai on a hand-made .bai as samtools writes it (bins with chunks, the metadata pseudo-bin, n_no_coor at the end), and
    build_bai -> read_bai on the reference-pinned multi-contig BAM: the first window of the first contig points at the first record"""
    import struct
    from longsom_amd import hostio
    raw = b"BAI\x01" + struct.pack("<i", 2)
    raw += struct.pack("<i", 2) + struct.pack("<Ii", 4681, 1) + struct.pack("<QQ", 100 << 16, 200 << 16) + struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", 1, 2, 3, 4)
    raw += struct.pack("<i", 3) + struct.pack("<QQQ", (100 << 16) | 7, (100 << 16) | 7, (150 << 16) | 9)
    raw += struct.pack("<i", 0) + struct.pack("<i", 0)
    raw += struct.pack("<Q", 5)
    p = tmp_path / "x.bai"
    p.write_bytes(raw)
    lin = hostio.read_bai(str(p))
    assert [v.tolist() for v in lin] == [[(100 << 16) | 7, (100 << 16) | 7, (150 << 16) | 9], []]
    import shutil
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    bam = str(tmp_path / "a.bam")
    shutil.copy(os.path.join(G, "pileup.rand.bam"), bam)
    lin = hostio.read_bai(hostio.build_bai(bam))
    names, lens, first = hostio.bam_header(bam)
    assert len(lin) == len(names) and [len(v) for v in lin] == [(int(l) + 16383) // 16384 for l in lens]
    assert int(lin[0][0]) == first                                   # block 0, offset of the first record
    flat = np.concatenate(lin)
    assert (np.diff(flat.astype(np.int64)) >= 0).all()              # a sorted file: the windows' first alignments come in file order


def test_bai_plan_regions_and_slices():
    lin = [np.array([(10 << 16) | 5, (10 << 16) | 5, (400 << 16) | 1, (900 << 16) | 0], np.uint64), np.zeros(0, np.uint64), np.array([(1500 << 16) | 3], np.uint64)]
    plan = regions.BaiPlan(lin, 3, 4, 2000)
    assert plan.bounds[0] == (0, 0) and plan.bounds[-1] == (3, 0) and plan.bounds == sorted(plan.bounds)
    assert plan.bounds[1] == (0, 3 * 16384) and plan.bounds[2] == (2, 0) and plan.bounds[3] == (2, 0)        # bytes 500, 1000, 1500 of 2000
    assert plan.start((0, 0)) == (10 << 16) | 5 and plan.start((0, 3 * 16384)) == 900 << 16
    assert plan.start((1, 0)) == (1500 << 16) | 3                   # nothing on contig 1: the next alignment in the file
    assert plan.start((2, 16384)) is None and plan.end((3, 0), 0) is None
    assert plan.end((0, 16384), 0) == (1500 << 16) | 3              # 4 windows further on contig 0 there is none: the next contig's first
    one = regions.BaiPlan(lin, 3, 1, 2000)
    assert one.bounds == [(0, 0), (3, 0)]


def test_the_guessed_end_of_a_slice_is_checked_on_the_host():
    """regions._last_key_of_block: the key of the last record that starts in the BGZF block a virtual offset points into (what ingest_slice
    looks at before it sends a slice to the device), against a walk over the whole inflated file"""
    import gzip
    import mmap
    import struct
    bam = os.path.join(G, "pileup.rand.bam")
    _, _, first = hostio.bam_header(bam)
    data = gzip.open(bam).read()
    with open(bam, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        coff, ubase, n_checked = 0, 0, 0
        u = first
        while coff < len(mm):
            end = regions._bgzf_block_end(mm, coff)
            isize = struct.unpack("<I", mm[end - 4:end])[0]
            if isize and u < ubase + isize:
                v = (coff << 16) | (u - ubase)                 # a record starts here: what a linear-index entry looks like
                last = None
                while u < ubase + isize:                       # (the walk over the whole file knows every record; the helper only what its block holds)
                    bs, tid, pos = struct.unpack_from("<Iii", data, u)
                    if u + 12 <= ubase + isize:
                        last = (1 << 62) if tid < 0 else (tid << 32) | max(pos, 0)
                    u += 4 + bs
                assert regions._last_key_of_block(mm, v) == last
                n_checked += 1
            coff, ubase = end, ubase + isize
        assert n_checked >= 2
        assert regions._last_key_of_block(mm, (7 << 16)) is None      # not a block: no guess, the device ingest decides


def test_an_index_older_than_its_bam_is_not_used(tmp_path, capsys):
    from longsom_amd import hostio
    bam, bai = tmp_path / "a.bam", tmp_path / "a.bam.bai"
    bai.write_bytes(b"BAI\x01\x00\x00\x00\x00"); bam.write_bytes(b"x")
    os.utime(bai, (1_000_000, 1_000_000)); os.utime(bam, (2_000_000, 2_000_000))
    assert hostio.find_bai(str(bam)) is None and "older" in capsys.readouterr().err
    os.utime(bai, (3_000_000, 3_000_000))
    assert hostio.find_bai(str(bam)) == str(bai)
    assert hostio.find_bai(str(tmp_path / "none.bam")) is None


# ---- the sharded per-cell genotyping (config 5 over several ranks), the CPU oracle standing in for each rank's device ----------------
class _OracleEngine:
    """what reanno.single_cell_genotype asks of an engine, answered by oracle/genotype_oracle.py over the reads a rank holds"""
    def __init__(self, rec, lens, celltype_of):
        self.rec, self.lens, self.celltype_of = rec, lens, celltype_of

    def genotype_cells_grouped(self, keys, alt_sym, group_off, params, max_depth):
        from oracle import genotype_oracle as go
        return go.genotype(self.rec, self.lens, self.celltype_of, keys, alt_sym, min_bq=params.min_bq, min_mq=params.min_mq, alt_only=params.alt_only,
                           strict_cb=params.strict_cb)

    def betabinom_sf4(self, k, n, alpha, beta):
        from scipy.stats import betabinom
        return np.asarray([int(round(round(float(betabinom.sf(int(a) - 0.001, int(b), alpha, beta)), 4) * 10000)) for a, b in zip(k, n)], np.int32)


def _geno_rank_main(rank, world, port, out_dir, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LSG_DIST_BACKEND="gloo")
    from longsom_amd import reanno
    comm = regions.Comm.from_env()
    bc = hostio.read_barcodes(os.path.join(G, "pileup.rand.barcodes.tsv"))
    dec = hostio.decode_bam(os.path.join(G, "pileup.rand.bam"), bc.barcodes, min_mapq=0)
    n_contigs = len(dec.contig_names)
    bounds = regions.balanced_boundaries(dec.records, n_contigs, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = dec.records.subset(regions.reads_overlapping(dec.records, lo, hi))         # the rank holds the reads that reach into its region, no others
    eng = _OracleEngine(mine, [int(x) for x in dec.contig_len], bc.celltype_of)
    out = os.path.join(out_dir, "geno.rank%d.tsv" % rank)
    stats = {}
    n = reanno.single_cell_genotype(eng, os.path.join(G, "pileup.rand.HCCV.tsv"), bc, dec.contig_names, out, alt_flag="All", min_bq=30, min_mq=60,
                                    comm=comm, region=(lo, hi), stats=stats)
    comm.barrier()
    q.put((rank, n, int(mine.n_reads), int(dec.records.n_reads), os.path.exists(out), sum(stats["covered"].values())))
    comm.close()


def test_sharded_genotyping_writes_the_reference_table(tmp_path):
    """two gloo ranks, each holding only its region's reads: the sites are genotyped where their reads are, one all-reduce places the rows,
    rank 0 writes the table the reference's HCCVSingleCellGenotype.py wrote for the whole BAM (tests/golden/pileup.rand.genotype.All.tsv)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_geno_rank_main, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = open(os.path.join(G, "pileup.rand.genotype.All.tsv")).read()
    assert open(tmp_path / "geno.rank0.tsv").read() == want
    assert res[0][4] and not res[1][4]                                   # only rank 0 writes
    assert res[0][1] == res[1][1] == want.count("\n") - 1
    assert all(0 < r[2] < r[3] for r in res)                             # each rank really held a part of the reads only
    assert res[0][5] == sum(1 for l in want.split("\n")[1:] if l and l.split("\t")[11] != ".")      # rows with coverage, as the re-annotation (on rank 0) counts them
    assert res[1][5] == 0                                                # (a rank that does not write builds no row)
