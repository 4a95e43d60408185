/*
 * synth_model.h — the synthetic long-read single-cell workload of BASELINE.md §4 / SURVEY.md §8d,
 * written once and compiled twice: by hipcc (device generator, synth.hip) and by g++ (host BAM
 * writer + host read-record generator, hostio/).  Integer-only, counter-based RNG: every draw is
 * hash(seed, read, position-or-slot, domain), so any position of any read can be produced
 * independently (lane-parallel on the GPU) and host and device agree bit for bit.
 *
 * Model (per read i):
 *   gene      reads [gene_read_off[g], gene_read_off[g+1]) belong to gene g (Zipf expression,
 *             allocation done on the host in longsom_amd/synth.py; genes sorted by position)
 *   span      aligned transcript length L ~ U[500,3000] clipped to the transcript, offset uniform
 *   flags     50 % reverse; 1 % supplementary, 0.5 % secondary, 0.2 % duplicate; MAPQ 92 % = 60
 *   barcode   2 % no CB tag, 3 % CB not in barcodes.tsv, else uniform over n_cb
 *   indels    transcript coordinates are cut into blocks of 8; a block lying inside one exon and
 *             inside the read carries at most one indel: deletion of d in [1,4] bases after offset 1
 *             (2.4 % of blocks = 0.3 %/base) or an insertion of 1-3 bases after offset 3 (1.6 %)
 *   bases     mismatch 0.5 %, 'N' call 0.02 %; germline het SNPs (all cells) and somatic SNVs (a
 *             clone of the cancer cells) decided by hashes of the site; quality 90 % U[20,60], 10 % U[2,19]
 * Events follow htslib/pysam pileup semantics: anchor base before an insertion -> I,
 * before a deletion -> D, interior deletion column -> O with the quality of the next query base.
 */
#ifndef LSG_SYNTH_MODEL_H
#define LSG_SYNTH_MODEL_H
#include <stdint.h>

#if defined(__HIPCC__)
#define LSG_HD __host__ __device__ inline
#else
#define LSG_HD static inline
#endif

#include "../../include/longsom_synth.h"   /* lsg_synth_model */

typedef struct {
    int32_t gene, tid, t_off, t_len, e0, e1, cb, clip5, clip3;
    uint16_t flag; uint8_t mapq;
} sm_read;

enum { SM_D_GENE = 1, SM_D_LEN, SM_D_OFF, SM_D_FLAG, SM_D_MAPQ, SM_D_CB, SM_D_BLOCK, SM_D_BASE, SM_D_QUAL,
       SM_D_INS, SM_D_CLIP, SM_D_SNP, SM_D_SNPREAD, SM_D_SNPCELL, SM_D_REF, SM_D_REFN, SM_D_BARCODE };

LSG_HD uint64_t sm_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
LSG_HD uint64_t sm_hash(uint64_t seed, uint64_t a, uint64_t b, uint64_t dom) {
    return sm_mix(sm_mix(sm_mix(seed ^ (dom * 0xD6E8FEB86659FD93ull)) ^ a) ^ b);
}
LSG_HD uint32_t sm_ppm(uint64_t h) { return (uint32_t)(h % 1000000ull); }

/* reference base (ASCII upper) of the synthetic genome */
LSG_HD uint8_t sm_ref_base(uint64_t seed, int32_t tid, int64_t pos) {
    if (sm_hash(seed, (uint64_t)tid, (uint64_t)(pos >> 8), SM_D_REFN) % 400ull == 0 && (pos & 255) < 100) return (uint8_t)'N';
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    return (uint8_t)acgt[sm_hash(seed, (uint64_t)tid, (uint64_t)pos, SM_D_REF) & 3ull];
}
/* symbol class (A,C,T,G = 0,1,2,3) of an ASCII base; 6 for N */
LSG_HD uint32_t sm_sym_of_base(uint8_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'T' ? 2u : b == 'G' ? 3u : 6u; }
LSG_HD uint8_t sm_base_of_sym(uint32_t s) { const char t[8] = {'A', 'C', 'T', 'G', '?', '?', 'N', '?'}; return (uint8_t)t[s & 7u]; }

LSG_HD void sm_read_header(const lsg_synth_model* m, int64_t i, sm_read* r) {
    int32_t lo = 0, hi = m->n_genes;             /* largest g with gene_read_off[g] <= i */
    const int64_t li = i - m->read_base;         /* i is the GLOBAL read index */
    while (hi - lo > 1) { int32_t mid = (lo + hi) >> 1; if (m->gene_read_off[mid] <= li) lo = mid; else hi = mid; }
    const int32_t g = lo;
    r->gene = g; r->tid = m->gene_tid[g];
    const int32_t x0 = m->gene_exon_off[g], x1 = m->gene_exon_off[g + 1];
    const int32_t tlen = m->exon_cum[x1 - 1] + m->exon_len[x1 - 1];
    int32_t L = 500 + (int32_t)(sm_hash(m->seed, (uint64_t)i, 0, SM_D_LEN) % 2501ull);
    if (L > tlen) L = tlen;
    r->t_len = L;
    r->t_off = (int32_t)(sm_hash(m->seed, (uint64_t)i, 0, SM_D_OFF) % (uint64_t)(tlen - L + 1));
    int32_t e = x0;                               /* exon containing transcript coordinate t_off */
    while (e + 1 < x1 && m->exon_cum[e + 1] <= r->t_off) ++e;
    r->e0 = e;
    const int32_t last = r->t_off + L - 1;
    while (e + 1 < x1 && m->exon_cum[e + 1] <= last) ++e;
    r->e1 = e;
    uint32_t f = 0;
    if (sm_hash(m->seed, (uint64_t)i, 0, SM_D_FLAG) & 1ull) f |= 0x10u;
    const uint32_t u = sm_ppm(sm_hash(m->seed, (uint64_t)i, 1, SM_D_FLAG));
    if (u < 10000u) f |= 0x800u; else if (u < 15000u) f |= 0x100u; else if (u < 17000u) f |= 0x400u;
    r->flag = (uint16_t)f;
    const uint32_t q = sm_ppm(sm_hash(m->seed, (uint64_t)i, 0, SM_D_MAPQ));
    r->mapq = q < 920000u ? (uint8_t)60 : (uint8_t)(sm_hash(m->seed, (uint64_t)i, 1, SM_D_MAPQ) % 60ull);
    const uint32_t c = sm_ppm(sm_hash(m->seed, (uint64_t)i, 0, SM_D_CB));
    if (c < 20000u) r->cb = -1; else if (c < 50000u) r->cb = -2;
    else r->cb = (int32_t)(sm_hash(m->seed, (uint64_t)i, 1, SM_D_CB) % (uint64_t)m->n_cb);
    r->clip5 = (int32_t)(sm_hash(m->seed, (uint64_t)i, 0, SM_D_CLIP) % 31ull);
    r->clip3 = (int32_t)(sm_hash(m->seed, (uint64_t)i, 1, SM_D_CLIP) % 31ull);
    if (sm_hash(m->seed, (uint64_t)i, 2, SM_D_CLIP) & 1ull) r->clip5 = 0;
    if (sm_hash(m->seed, (uint64_t)i, 3, SM_D_CLIP) & 1ull) r->clip3 = 0;
}

/* Segment k of a read = its part of exon e0 + k: reference start and length. */
LSG_HD void sm_read_segment(const lsg_synth_model* m, const sm_read* r, int32_t k, int32_t* start, int32_t* len) {
    const int32_t x = r->e0 + k, xt0 = m->exon_cum[x], xt1 = xt0 + m->exon_len[x], end = r->t_off + r->t_len;
    const int32_t lo = r->t_off > xt0 ? r->t_off : xt0, hi = end < xt1 ? end : xt1;
    *start = m->exon_start[x] + (lo - xt0); *len = hi - lo;
}
/* LSG_LAYOUT_PHASED (include/longsom_hip.h): the place of a segment that starts at reference position `start`, behind `cur` events of its
 * read's region (the region itself starts at a multiple of 128): the next offset congruent to start modulo 128. */
LSG_HD int64_t sm_phase_place(int64_t cur, int32_t start) { return cur + (((int64_t)start - cur) & 127); }
/* Events of a read's region under the model's layout: compact = its t_len events; phased = its segments at their phases, rounded up to 128. */
LSG_HD int64_t sm_read_region(const lsg_synth_model* m, const sm_read* r) {
    if (m->layout != LSG_LAYOUT_PHASED) return r->t_len;
    int64_t cur = 0;
    for (int32_t k = 0; k <= r->e1 - r->e0; ++k) { int32_t st, ln; sm_read_segment(m, r, k, &st, &ln); cur = sm_phase_place(cur, st) + ln; }
    return (cur + 127) & ~(int64_t)127;
}

/* indel carried by block blk (transcript coords [8blk, 8blk+8)) of read i: 0 none, >0 deletion of
 * that many bases after offset 1, <0 insertion of that many bases after offset 3.  [lo,hi) is the
 * transcript interval in which the block must lie entirely (read span intersected with the exon). */
LSG_HD int32_t sm_block_indel(const lsg_synth_model* m, int64_t i, int32_t blk, int32_t lo, int32_t hi) {
    if (blk * 8 < lo || blk * 8 + 8 > hi) return 0;
    const uint64_t h = sm_hash(m->seed, (uint64_t)i, (uint64_t)blk, SM_D_BLOCK);
    const uint32_t u = sm_ppm(h);
    if (u < 24000u) return 1 + (int32_t)((h >> 40) & 3ull);
    if (u < 40000u) return -(1 + (int32_t)((h >> 40) % 3ull));
    return 0;
}
LSG_HD uint32_t sm_qual(const lsg_synth_model* m, int64_t i, int32_t slot) {
    const uint64_t h = sm_hash(m->seed, (uint64_t)i, (uint64_t)(uint32_t)slot, SM_D_QUAL);
    return sm_ppm(h) < 900000u ? 20u + (uint32_t)((h >> 40) % 41ull) : 2u + (uint32_t)((h >> 40) % 18ull);
}
/* base call (symbol class 0..3 or 6 = N) of read i at transcript coordinate j = reference (tid,pos) */
LSG_HD uint32_t sm_base_call(const lsg_synth_model* m, int64_t i, const sm_read* r, int32_t j, int64_t pos) {
    const uint8_t rb = sm_ref_base(m->seed, r->tid, pos);
    uint32_t rs = sm_sym_of_base(rb);
    const uint64_t h = sm_hash(m->seed, (uint64_t)i, (uint64_t)(uint32_t)j, SM_D_BASE);
    const uint32_t u = sm_ppm(h);
    if (u < 200u) return 6u;
    if (rs == 6u) rs = (uint32_t)((h >> 44) & 3ull);           /* reference N: any call */
    /* planted SNVs */
    const uint64_t sh = sm_hash(m->seed, (uint64_t)r->tid, (uint64_t)pos, SM_D_SNP);
    const uint64_t cls = sh % ((uint64_t)m->snp_mod * 8ull);
    const uint32_t alt = (rs + 1u + (uint32_t)((sh >> 40) % 3ull)) & 3u;
    if (cls < 8ull) {                                            /* germline het: every cell, half of the reads */
        if (sm_hash(m->seed, (uint64_t)i, (uint64_t)pos, SM_D_SNPREAD) & 1ull) return alt;
    } else if (cls == 8ull && r->cb >= 0 && m->celltype_of[r->cb] == 0) {   /* somatic, cancer clone */
        const uint32_t frac = 300000u + (uint32_t)((sh >> 20) % 700001ull);
        if (sm_ppm(sm_hash(m->seed, (uint64_t)r->cb, (uint64_t)pos ^ ((uint64_t)r->tid << 40), SM_D_SNPCELL)) < frac &&
            (sm_hash(m->seed, (uint64_t)i, (uint64_t)pos, SM_D_SNPREAD) & 1ull))
            return alt;
    }
    if (u < 200u + 5000u) return (rs + 1u + (uint32_t)((h >> 40) % 3ull)) & 3u;   /* sequencing mismatch */
    return rs;
}

/* Pileup event of read i at transcript coordinate j inside exon x (transcript interval
 * [xt0, xt1), reference start xs).  Returns LSG_EVENT(sym, qual). */
LSG_HD uint16_t sm_event(const lsg_synth_model* m, int64_t i, const sm_read* r, int32_t j, int32_t xt0, int32_t xt1, int32_t xs) {
    int32_t lo = r->t_off > xt0 ? r->t_off : xt0;
    int32_t hi = r->t_off + r->t_len < xt1 ? r->t_off + r->t_len : xt1;
    const int32_t blk = j >> 3, k = j & 7;
    const int32_t ind = sm_block_indel(m, i, blk, lo, hi);
    uint32_t sym, q;
    if (ind > 0 && k >= 2 && k <= 1 + ind) {              /* interior deletion column: quality of next base */
        sym = 7u; q = sm_qual(m, i, blk * 8 + 2 + ind);
    } else {
        q = sm_qual(m, i, j);
        if (ind > 0 && k == 1) sym = 5u;                  /* anchor before a deletion */
        else if (ind < 0 && k == 3) sym = 4u;             /* anchor before an insertion */
        else sym = sm_base_call(m, i, r, j, (int64_t)xs + (j - xt0));
    }
    return LSG_EVENT(sym, q);
}

/* 16-mer barcode string of dense id cb (cb >= 0), or of an unlisted barcode (cb == -2, keyed by read) */
LSG_HD void sm_barcode(uint64_t seed, int64_t cb_or_read, int unlisted, char out[17]) {
    uint64_t h = sm_hash(seed, (uint64_t)cb_or_read, (uint64_t)unlisted, SM_D_BARCODE);
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    /* listed barcodes encode their id in base 4 in the first 12 letters (unique), unlisted start with 'N' */
    uint64_t id = (uint64_t)cb_or_read;
    for (int k = 0; k < 16; ++k) {
        if (k < 12 && !unlisted) { out[k] = acgt[id & 3ull]; id >>= 2; }
        else { out[k] = acgt[h & 3ull]; h >>= 2; }
    }
    if (unlisted) out[0] = 'N';
    out[16] = 0;
}
#endif
