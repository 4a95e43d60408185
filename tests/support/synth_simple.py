"""Small random read-record generator (numpy) for unit tests of the count kernels.

This is NOT the BASELINE workload model (that is longsom_amd/synth.py, mirrored in HIP); it only
produces structurally valid ReadRecords with controllable depth / barcode skew so that the wave
kernel, the deep kernel's staged passes and its stream mode are all exercised.
"""
import numpy as np

from longsom_amd.engine import ReadRecords


def random_reference(rng, length, n_frac=0.01):
    ref = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
    n_runs = max(1, int(length * n_frac / 50))
    for _ in range(n_runs):
        s = int(rng.integers(0, max(1, length - 50)))
        ref[s:s + int(rng.integers(1, 50))] = ord("N")
    return ref


def random_records(seed, n_reads, contig_lens, n_cb, hot_regions=(), hot_frac=0.0, cb_skew=0.0,
                   max_exons=4, exon_len=(30, 400), intron_len=(20, 3000)):
    """hot_regions: list of (tid, start, end); a fraction hot_frac of the reads start inside one.
    cb_skew: probability that a read takes barcode 0 (forces huge single-barcode runs)."""
    rng = np.random.default_rng(seed)
    contig_lens = np.asarray(contig_lens, dtype=np.int64)
    read_tid = np.zeros(n_reads, np.int32); read_pos = np.zeros(n_reads, np.int32)
    read_flag = np.zeros(n_reads, np.uint16); read_mapq = np.zeros(n_reads, np.uint8)
    read_cb = np.zeros(n_reads, np.int32)
    seg_read, seg_start, seg_len = [], [], []
    for r in range(n_reads):
        if hot_regions and rng.random() < hot_frac:
            tid, hs, he = hot_regions[int(rng.integers(0, len(hot_regions)))]
            start = int(rng.integers(max(0, hs - 20), he))
        else:
            tid = int(rng.integers(0, len(contig_lens)))
            start = int(rng.integers(0, max(1, contig_lens[tid] - 50)))
        clen = int(contig_lens[tid])
        pos = start
        n_ex = int(rng.integers(1, max_exons + 1))
        first = True
        for _ in range(n_ex):
            ln = int(rng.integers(exon_len[0], exon_len[1]))
            if pos >= clen:
                break
            ln = min(ln, clen - pos)
            if ln <= 0:
                break
            seg_read.append(r); seg_start.append(pos); seg_len.append(ln)
            if first:
                read_pos[r] = pos; first = False
            pos += ln + int(rng.integers(intron_len[0], intron_len[1]))
        read_tid[r] = tid
        f = 0
        u = rng.random()
        if u < 0.5: f |= 0x10
        u = rng.random()
        if u < 0.01: f |= 0x800
        elif u < 0.015: f |= 0x100
        elif u < 0.017: f |= 0x400
        elif u < 0.018: f |= 0x200
        elif u < 0.019: f |= 0x4
        elif u < 0.022: f |= 0x1          # paired, not proper pair -> orphan
        elif u < 0.025: f |= 0x3          # proper pair -> kept
        read_flag[r] = f
        read_mapq[r] = 60 if rng.random() < 0.92 else int(rng.integers(0, 60))
        u = rng.random()
        if u < 0.02: read_cb[r] = -1
        elif rng.random() < cb_skew: read_cb[r] = 0
        else: read_cb[r] = int(rng.integers(0, n_cb))
    seg_read = np.asarray(seg_read, np.uint32); seg_start = np.asarray(seg_start, np.int32); seg_len = np.asarray(seg_len, np.int32)
    n_ev = int(seg_len.sum())
    seg_ev_off = np.concatenate([[0], np.cumsum(seg_len)[:-1]]).astype(np.int64) if len(seg_len) else np.zeros(0, np.int64)
    # symbols: mostly A/C/T/G, some indel anchors / deletions / N / NA
    sym = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 15], dtype=np.uint16), size=n_ev,
                     p=[0.235, 0.235, 0.235, 0.235, 0.01, 0.01, 0.005, 0.025, 0.01])
    qual = np.where(rng.random(n_ev) < 0.9, rng.integers(20, 61, n_ev), rng.integers(2, 20, n_ev)).astype(np.uint16)
    events = np.where(sym < 8, 0x0800 | (sym << 8) | qual, 0).astype(np.uint16)      # LSG_EVENT(sym, qual)
    return ReadRecords(read_tid, read_pos, read_flag, read_mapq, read_cb, seg_read, seg_start, seg_len, seg_ev_off, events)
