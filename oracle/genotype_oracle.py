"""TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the per-cell genotyping of
/root/reference/workflow/scripts/CellTypeReannotation/HCCVSingleCellGenotype.py:82-220 (twin: SNVCalling/SingleCellGenotype.py)
on the read-record arrays, plus the text of its output rows.  Only tests/ may import it.

PARITY UNPINNED for the pileup part: the reference's arithmetic here IS pysam's pileup (absent from this image and from
the reference tree), so it is restated from the htslib/pysam semantics of SURVEY.md §8a exactly as oracle/count_oracle.c
does, and pinned by hand-derived known answers (tests/test_genotype_cpu.py).  The per-cell status / p-value / text part
uses scipy.stats.betabinom like the reference (:204).
"""
import numpy as np


def depth_cap_drops(rec, tid, start, end, min_mq, flag_exclude, ignore_orphans, max_depth):
    """reads the pileup of (tid, [start, end)) with max_depth drops (HCCVSingleCellGenotype.py:122; htslib bam_plp_push): the reads that
    overlap the region arrive in file (coordinate) order; the first read of a start position always enters the buffer, a later one of
    the same position is dropped while (reads that entered and end at or after that position) + 1 > max_depth.  The pileup's own read
    filter (flag filter without the supplementary bit, which only :168 tests later; ignore_orphans; min_mapping_quality) comes first."""
    R = rec.n_reads
    ends = rec.read_pos.astype(np.int64) + 1
    for s in range(rec.n_segs):                               # a read's last pileup column + 1 (its last segment's end; pos + 1 without segments)
        ends[int(rec.seg_read[s])] = int(rec.seg_start[s]) + int(rec.seg_len[s])
    order = sorted(range(R), key=lambda i: (int(rec.read_tid[i]), int(rec.read_pos[i])))
    pool = flag_exclude & ~0x800
    dropped, buffered, cur, first = set(), [], None, True
    for i in order:
        if int(rec.read_tid[i]) != tid or int(rec.read_pos[i]) >= end or ends[i] <= start:
            continue
        f = int(rec.read_flag[i])
        if int(rec.read_mapq[i]) < min_mq or (f & pool) or (ignore_orphans and (f & 1) and not (f & 2)):
            continue
        p = int(rec.read_pos[i])
        if p != cur:
            cur, first = p, True
            buffered = [e for e in buffered if e >= p]
        if not first and len(buffered) + 1 > max_depth:
            dropped.add(i)
            continue
        first = False
        buffered.append(int(ends[i]))
    return dropped


def genotype(rec, contig_len, celltype_of, site_keys, alt_sym, min_bq=30, min_mq=60, flag_exclude=0xF04, ignore_orphans=1,
             alt_only=0, strict_cb=1, group_off=None, max_depth=0):
    """-> (dp, alt) uint32 [n_sites, n_cb].  Plain loops over segments (small inputs only).  group_off + max_depth: the sites
    [group_off[g], group_off[g + 1]) are one pileup call over [first site - 1, last site + 1) with that depth cap (:109-122)."""
    n_cb = len(celltype_of)
    site_keys = np.asarray(site_keys, np.int64)
    dp = np.zeros((len(site_keys), n_cb), np.uint32)
    alt = np.zeros((len(site_keys), n_cb), np.uint32)
    dropped_at = [set()] * len(site_keys)
    if group_off is not None and max_depth > 0:
        for g in range(len(group_off) - 1):
            a, b = int(group_off[g]), int(group_off[g + 1])
            if b > a:
                d = depth_cap_drops(rec, int(site_keys[a]) >> 32, (int(site_keys[a]) & 0xFFFFFFFF) - 1, (int(site_keys[b - 1]) & 0xFFFFFFFF) + 1,
                                    min_mq, flag_exclude, ignore_orphans, max_depth)
                for i in range(a, b):
                    dropped_at[i] = d
    for s in range(rec.n_segs):
        r = int(rec.seg_read[s])
        flag = int(rec.read_flag[r]); cb = int(rec.read_cb[r]); tid = int(rec.read_tid[r])
        if flag & flag_exclude:                                    # pileup flag filter + :168
            continue
        if int(rec.read_mapq[r]) < min_mq:                         # min_mapping_quality (:123)
            continue
        if ignore_orphans and (flag & 0x1) and not (flag & 0x2):   # pysam pileup default
            continue
        if cb < 0 or cb >= n_cb or celltype_of[cb] == 255:         # no CB / not in barcodes.tsv (:160-164)
            continue
        if strict_cb and (flag & 0x8000):                          # raw CB "XXXX-1" is not a key of the cleaned table (:160-161)
            continue
        if tid < 0 or tid >= len(contig_len):
            continue
        st, ln = int(rec.seg_start[s]), int(rec.seg_len[s])
        if st < 0 or ln <= 0 or st + ln > int(contig_len[tid]):
            continue
        k_lo = (tid << 32) | st
        i = int(np.searchsorted(site_keys, k_lo, side="left"))
        while i < len(site_keys) and site_keys[i] < k_lo + ln:
            ev = int(rec.events[int(rec.seg_ev_off[s]) + int(site_keys[i]) - k_lo])
            i += 1
            if r in dropped_at[i - 1]:                                # the pileup of this site's window never saw the read
                continue
            if not (ev & 0x0800) or (ev & 0xff) < min_bq:          # 'NA'; pileup min_base_quality (:123)
                continue
            sym = (ev >> 8) & 7
            if sym > 6:                                            # 'O' is not in Bases (:148)
                continue
            is_alt = sym == int(alt_sym[i - 1])
            if alt_only and not is_alt:                            # --alt_flag Alt (:150-151)
                continue
            dp[i - 1, cb] += 1
            if is_alt:
                alt[i - 1, cb] += 1
    return dp, alt


def cell_row(chrom, pos0, ref_exp, alt_exp, ctype_exp, ncells_exp, bc, ctype, DP, ALT, alpha2, beta2, pval, chrm_conta):
    """One output line of run_interval (:181-216), without the newline."""
    from scipy.stats import betabinom
    VAF = '.'; BETABIN = '.'; MUTATED = 'NoCoverage'
    if DP > 0:
        if ALT > 0:
            VAF = round(ALT / DP, 4)
            if chrm_conta == 'True' and str(chrom) == 'chrM':
                MUTATED = 'LowVAFChrM' if VAF < 0.3 else 'PASS'
            else:
                BETABIN = round(betabinom.sf(ALT - 0.001, DP, alpha2, beta2), 4)
                MUTATED = 'PASS' if BETABIN < pval else 'BetaBin_problem'
        else:
            VAF = float(0)
            MUTATED = 'NoAltReads'
    return '\t'.join([str(chrom), str(pos0 + 1), str(pos0 + 1), ref_exp, alt_exp, str(ctype_exp), str(ncells_exp), bc, ctype,
                      str(DP), str(ALT), str(VAF), str(BETABIN), str(MUTATED)])
