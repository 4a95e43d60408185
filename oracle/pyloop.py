"""cpu_pyloop — TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg; never on the product path).

A per-read Python loop with the cost structure of the reference's run_interval
(/root/reference/workflow/scripts/SNVCalling/BaseCellCounter.py:198-312): for every pileup column it holds the reference's
per-column lists (names, qualities, pysam-style sequence strings), runs EasyReadPileup's string tests (:152-180) and the per-read
loop (:236-279: dict counters, barcode lists), takes len(set(...)) per symbol and for the column (:283,:292), and formats the row
text (:297-309).  The columns come from the decoded read-record arrays (numpy), so what pysam spends on building PileupRead /
AlignedSegment objects per read per column (:225) is NOT included: this is a LOWER bound on the reference's own time per column.
It is the closest legal stand-in for "CPU BaseCellCounter" on the GPU box (the reference needs pysam, which is not installed and
may not travel), and its rows are checked against oracle/count_oracle.c so that the timed work is the real work.

windows() cuts a contig into the reference's 50 kb windows (MakeWindows, :81-113) so that the all-cores variant can fan them out
over a process pool exactly as the reference does (:392-402)."""
import numpy as np

SYMS = ["A", "C", "T", "G", "I", "D", "N", "O"]          # event class -> EasyReadPileup's symbol
PYSAM_STR = ["A", "C", "T", "G", "A+1A", "A-1N", "N", "*"]   # a pysam string of that class (what get_query_sequences would hand over)
ALLELES = ["A", "C", "T", "G", "I", "D"]


def easy_read_pileup(lst, ref_base):
    """EasyReadPileup (:152-180), statement by statement"""
    bases = set(["A", "C", "T", "G", "N"])
    ac = 0
    new = []
    for x in lst:
        ln = len(x)
        up = x.upper()
        if up in bases:
            new.append(up)
            if up != ref_base:
                ac = ac + 1
        elif ln > 1 and x[1] == "-":
            new.append("D"); ac = ac + 1
        elif ln > 1 and x[1] == "+":
            new.append("I"); ac = ac + 1
        elif x == "*":
            new.append("O")
        else:
            new.append("NA")
    return new, ac


def columns_of(rec, tid, lo, hi, min_bq, admitted):
    """per column of [lo, hi) on contig tid: (pos, [(read index, class, quality)]) — the pileup iterator's part, in numpy"""
    sel = np.flatnonzero((rec.read_tid[rec.seg_read] == tid) & admitted[rec.seg_read] & (rec.seg_start < hi) & (rec.seg_start + rec.seg_len > lo))
    pos_l, read_l, ev_l = [], [], []
    for s in sel.tolist():
        st, ln, off = int(rec.seg_start[s]), int(rec.seg_len[s]), int(rec.seg_ev_off[s])
        a, b = max(st, lo), min(st + ln, hi)
        pos_l.append(np.arange(a, b, dtype=np.int64)); read_l.append(np.full(b - a, rec.seg_read[s], np.int64)); ev_l.append(rec.events[off + a - st: off + b - st])
    if not pos_l:
        return
    pos, rd, ev = np.concatenate(pos_l), np.concatenate(read_l), np.concatenate(ev_l).astype(np.int64)
    keep = ((ev & 0x800) != 0) & ((ev & 0xff) >= min_bq)                 # 'NA' entries and the base-quality skip
    pos, rd, ev = pos[keep], rd[keep], ev[keep]
    order = np.argsort(pos, kind="stable")
    pos, rd, ev = pos[order], rd[order], ev[order]
    cut = np.flatnonzero(np.diff(pos)) + 1
    for a, b in zip(np.concatenate([[0], cut]), np.concatenate([cut, [len(pos)]])):
        yield int(pos[a]), rd[a:b].tolist(), ((ev[a:b] >> 8) & 7).tolist(), (ev[a:b] & 0xff).tolist()


def run_interval(rec, barcodes, ref, chrom, tid, lo, hi, admitted, min_cov=5, min_cc=5, min_bq=20):
    """run_interval (:182-320) over [lo, hi): the list of output lines"""
    positions = []
    rev = ((rec.read_flag >> 4) & 1).tolist()
    cb = rec.read_cb.tolist()
    for pos, reads, classes, quals in columns_of(rec, tid, lo, hi, min_bq, admitted):
        if pos < 1:
            continue
        ref_base = chr(ref[pos]).upper()
        dp = len(reads)
        cells = []
        if dp >= min_cov and ref_base != "N":
            pileup_list = [PYSAM_STR[c].lower() if rev[r] else PYSAM_STR[c] for r, c in zip(reads, classes)]
            new_list, ac = easy_read_pileup(pileup_list, ref_base)
            callable_sites = [x for x in new_list if x != "NA"]
            if len(callable_sites) < min_cov or ac < 0:
                continue
            base_counts = {"A": 0, "C": 0, "T": 0, "G": 0, "D": 0, "I": 0, "N": 0, "O": 0}
            base_quals = {"A": 0, "C": 0, "T": 0, "G": 0, "D": 0, "I": 0, "N": 0, "O": 0}
            cell_counts = {"A": [], "C": [], "T": [], "G": [], "D": [], "I": [], "N": [], "O": []}
            counts_f = {"A": 0, "C": 0, "T": 0, "G": 0, "D": 0, "I": 0, "N": 0, "O": 0}
            counts_r = {"A": 0, "C": 0, "T": 0, "G": 0, "D": 0, "I": 0, "N": 0, "O": 0}
            count = 0
            for i in range(0, len(reads)):
                r = reads[i]
                barcode = barcodes[cb[r]]
                barcode = barcode.split("-")[0]
                base = new_list[i]
                bq = quals[i]
                if base in base_counts.keys():
                    count = count + 1
                    base_counts[base] = base_counts[base] + 1
                    base_quals[base] = base_quals[base] + bq
                    if rev[r]:
                        counts_r[base] = counts_r[base] + 1
                    else:
                        counts_f[base] = counts_f[base] + 1
                    cell_counts[base].append(barcode)
                    cells.append(barcode)
            if count >= min_cov:
                cc2 = {x: len(set(cell_counts[x])) for x in cell_counts.keys()}
                nc = len(set(cells))
                if nc >= min_cc:
                    info = "|".join(["DP", "NC", "CC", "BC", "BQ", "BCf", "BCr"])
                    j = lambda d: ":".join([str(d[x]) for x in ALLELES])
                    line = "\t".join([str(chrom), str(pos + 1), ref_base, info,
                                      "|".join([str(count), str(nc), j(cc2), j(base_counts), j(base_quals), j(counts_f), j(counts_r)])])
                    positions.append(line)
    return positions


def windows(contig_len, size=50000):
    """MakeWindows (:81-113): [1, 50001), [50001, ...) per contig"""
    out = []
    for tid, ln in enumerate(contig_len):
        x = 1
        while x < ln:
            out.append((tid, x, min(x + size, int(ln)))); x += size
    return out


def admitted_reads(rec, celltype_of, ct, min_mq=60, flag_exclude=0xF04):
    ok = ((rec.read_flag & flag_exclude) == 0) & (rec.read_mapq >= min_mq) & (rec.read_cb >= 0)
    ok &= ~(((rec.read_flag & 1) != 0) & ((rec.read_flag & 2) == 0))
    ct_of = np.where(rec.read_cb >= 0, np.asarray(celltype_of)[np.maximum(rec.read_cb, 0)], 255)
    return ok & (ct_of == ct)


_G = {}


def _pool_init(rec, barcodes, refs, names, celltype_of):
    _G.update(rec=rec, barcodes=barcodes, refs=refs, names=names, adm=[admitted_reads(rec, celltype_of, ct) for ct in range(2)])


def _pool_work(job):
    ct, tid, lo, hi = job
    return len(run_interval(_G["rec"], _G["barcodes"], _G["refs"][tid], _G["names"][tid], tid, lo, hi, _G["adm"][ct]))


def count_windows(rec, barcodes, refs, names, celltype_of, jobs, nprocs=1):
    """rows written for the (cell type, window) jobs; nprocs > 1: a process pool over the jobs, as the reference's mp.Pool(CORE)"""
    if nprocs <= 1:
        _pool_init(rec, barcodes, refs, names, celltype_of)
        return sum(_pool_work(j) for j in jobs)
    import multiprocessing as mp
    with mp.get_context("fork").Pool(nprocs, initializer=_pool_init, initargs=(rec, barcodes, refs, names, celltype_of)) as pool:
        return sum(pool.map(_pool_work, jobs, chunksize=1))
