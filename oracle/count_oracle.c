/*
 * count_oracle.c — TEST INFRASTRUCTURE ONLY (CPU restatement, never shipped or called by the
 * product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * Column-major restatement of the counting loop of the reference's BaseCellCounter
 * (/root/reference/workflow/scripts/SNVCalling/BaseCellCounter.py:198-312) working from the
 * decoded read-record arrays (the same arrays the HIP path consumes; the decode itself is checked
 * against plp_oracle.c, which walks CIGARs the way htslib's bam_plp does).
 *
 * Structure mirrors the reference: build, for every pileup column, the list of entries
 * (read, symbol, quality) — here by sorting the expanded entries by (contig, position) — then per
 * column run the per-read loop (:236-279): count, BASE_COUNTS, BASE_QUALITIES, strand counts,
 * CELL_COUNTS lists and the CELLS list; the distinct-cell numbers are len(set(...)) (:283,:292).
 *
 * Parity status: pinned by tests/golden/kat_*.json (hand-derived known answers from the htslib /
 * pysam semantics in SURVEY.md §8a) — the reference's pileup cannot run here (no pysam), so the
 * pileup stage is "parity unpinned" against the reference itself; see DESIGN.md §6.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t key;      /* (tid << 32) | pos */
    int32_t cb;
    uint8_t sym, qual, rev, pad;
} entry_t;

static int cmp_entry(const void* a, const void* b) {
    const entry_t* x = (const entry_t*)a; const entry_t* y = (const entry_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->cb != y->cb) return x->cb < y->cb ? -1 : 1;
    return 0;
}

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : (x > y);
}

/* number of distinct values in v[0..n) (len(set(v))) */
static int distinct(int32_t* v, int n) {
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof(int32_t), cmp_i32);
    int d = 1;
    for (int i = 1; i < n; ++i) d += v[i] != v[i - 1];
    return d;
}

/*
 * Returns the number of emitted rows of cell type `ct` (rows are written in (tid,pos) order up to
 * `capacity`), or -1 on allocation failure.  *n_columns receives the number of columns with >= 1
 * counted entry.  Row layout = 42 words: DP, NC, CC[8], BC[8], BQ[8], BCf[8], BCr[8], classes in
 * the order A,C,T,G,I,D,N,O (BaseCellCalling.step1.py:20).
 */
int64_t lso_count(int64_t n_reads, int64_t n_segs,
                  const int32_t* read_tid, const uint16_t* read_flag, const uint8_t* read_mapq, const int32_t* read_cb,
                  const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, const int64_t* seg_ev_off,
                  const uint16_t* events,
                  int32_t n_contigs, const int64_t* contig_len, const uint8_t* const* ref,
                  const uint8_t* celltype_of, int32_t n_cb, int32_t ct,
                  int32_t min_bq, int32_t min_mq, int32_t min_dp, int32_t min_cc, uint32_t flag_exclude, int32_t ignore_orphans,
                  int64_t* out_keys, uint8_t* out_ref, uint32_t* out_counts, int64_t capacity, int64_t* n_columns)
{
    (void)n_reads;
    /* pass 1: how many entries survive read admission + the base-quality gate */
    int64_t n_ent = 0;
    entry_t* ents = NULL;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            ents = (entry_t*)malloc(sizeof(entry_t) * (size_t)(n_ent > 0 ? n_ent : 1));
            if (!ents) return -1;
        }
        int64_t w = 0;
        for (int64_t s = 0; s < n_segs; ++s) {
            uint32_t r = seg_read[s];
            uint32_t flag = read_flag[r];
            /* read admission: pysam pileup flag_filter + min_mapping_quality + ignore_orphans
               (BaseCellCounter.py:191), is_secondary/is_duplicate/is_supplementary (:249),
               CB present (:240-243), barcode belongs to this cell type's BAM
               (SplitBamCellTypes.py:83-90,110-113,173) */
            if (flag & flag_exclude) continue;
            if ((int)read_mapq[r] < min_mq) continue;
            if (ignore_orphans && (flag & 1u) && !(flag & 2u)) continue;
            int32_t cb = read_cb[r];
            if (cb < 0 || cb >= n_cb) continue;
            if ((int)celltype_of[cb] != ct) continue;
            int32_t tid = read_tid[r];
            if (tid < 0 || tid >= n_contigs) continue;
            int32_t st = seg_start[s], ln = seg_len[s];
            if (st < 0 || ln <= 0 || (int64_t)st + ln > contig_len[tid]) continue;
            for (int32_t i = 0; i < ln; ++i) {
                uint16_t ev = events[seg_ev_off[s] + i];
                int q = ev & 0xff, sym = (ev >> 8) & 7;   /* LSG_EVENT: 0x0800 | class << 8 | qual, 0 = 'NA' */
                if (!(ev & 0x0800)) continue;      /* 'NA' symbols: not in BASE_COUNTS.keys() (:258) */
                if (q < min_bq) continue;          /* pileup_base_qual_skip: every accessor drops it */
                if (pass == 1) {
                    entry_t* e = &ents[w];
                    e->key = ((int64_t)tid << 32) | (int64_t)(st + i);
                    e->cb = cb; e->sym = (uint8_t)sym; e->qual = (uint8_t)q; e->rev = (uint8_t)((flag >> 4) & 1u); e->pad = 0;
                }
                ++w;
            }
        }
        if (pass == 0) { n_ent = w; continue; }

        qsort(ents, (size_t)n_ent, sizeof(entry_t), cmp_entry);
        int64_t n_rows = 0, n_cols = 0;
        int32_t* cells = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_ent > 0 ? n_ent : 1));
        if (!cells) { free(ents); return -1; }
        int64_t i = 0;
        while (i < n_ent) {
            int64_t j = i;
            while (j < n_ent && ents[j].key == ents[i].key) ++j;
            int32_t tid = (int32_t)(ents[i].key >> 32);
            int64_t pos = ents[i].key & 0xffffffffll;
            /* windows start at 1: 0-based position 0 is never visited (MakeWindows, :86) */
            if (pos >= 1) {
                ++n_cols;
                uint32_t bc[8] = {0}, bq[8] = {0}, bcf[8] = {0}, bcr[8] = {0}, cc[8] = {0};
                uint32_t count = 0;
                for (int64_t k = i; k < j; ++k) {              /* per-read loop :236-279 */
                    const entry_t* e = &ents[k];
                    ++count; ++bc[e->sym]; bq[e->sym] += e->qual;
                    if (e->rev) ++bcr[e->sym]; else ++bcf[e->sym];
                }
                for (int sym = 0; sym < 8; ++sym) {            /* CELL_COUNTS2 :283 */
                    int m = 0;
                    for (int64_t k = i; k < j; ++k) if (ents[k].sym == sym) cells[m++] = ents[k].cb;
                    cc[sym] = (uint32_t)distinct(cells, m);
                }
                int m = 0;
                for (int64_t k = i; k < j; ++k) cells[m++] = ents[k].cb;
                uint32_t nc = (uint32_t)distinct(cells, m);   /* len(set(CELLS)) :292 */
                uint8_t refb = ref && ref[tid] ? ref[tid][pos] : (uint8_t)'?';
                /* gates :211 (ref != 'N'), :282 (count >= MIN_COV), :294 (NC >= MIN_CC) */
                if (refb != 'N' && (int)count >= min_dp && (int)nc >= min_cc) {
                    if (n_rows < capacity) {
                        uint32_t* o = out_counts + n_rows * 42;
                        out_keys[n_rows] = ents[i].key; out_ref[n_rows] = refb;
                        o[0] = count; o[1] = nc;
                        for (int sym = 0; sym < 8; ++sym) {
                            o[2 + sym] = cc[sym]; o[10 + sym] = bc[sym]; o[18 + sym] = bq[sym];
                            o[26 + sym] = bcf[sym]; o[34 + sym] = bcr[sym];
                        }
                    }
                    ++n_rows;
                }
            }
            i = j;
        }
        free(cells); free(ents);
        if (n_columns) *n_columns = n_cols;
        return n_rows;
    }
    return -1;
}
