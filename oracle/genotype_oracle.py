"""TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the per-cell genotyping of
/root/reference/workflow/scripts/CellTypeReannotation/HCCVSingleCellGenotype.py:82-220 (twin: SNVCalling/SingleCellGenotype.py)
on the read-record arrays, plus the text of its output rows.  Only tests/ may import it.

PARITY UNPINNED for the pileup part: the reference's arithmetic here IS pysam's pileup (absent from this image and from
the reference tree), so it is restated from the htslib/pysam semantics of SURVEY.md §8a exactly as oracle/count_oracle.c
does, and pinned by hand-derived known answers (tests/test_genotype_cpu.py).  The per-cell status / p-value / text part
uses scipy.stats.betabinom like the reference (:204).
"""
import numpy as np


def genotype(rec, contig_len, celltype_of, site_keys, alt_sym, min_bq=30, min_mq=60, flag_exclude=0xF04, ignore_orphans=1,
             alt_only=0, strict_cb=1):
    """-> (dp, alt) uint32 [n_sites, n_cb].  Plain loops over segments (small inputs only)."""
    n_cb = len(celltype_of)
    site_keys = np.asarray(site_keys, np.int64)
    dp = np.zeros((len(site_keys), n_cb), np.uint32)
    alt = np.zeros((len(site_keys), n_cb), np.uint32)
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
