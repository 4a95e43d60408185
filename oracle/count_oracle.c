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

/* ------------------------------------------------------------------------------------------------
 * The per-column loop (:198-312) over entries already sorted by (key, cb): shared by the single-threaded
 * lso_count below and by the region-parallel lso_count_mt.  Rows are appended to (keys, refb, counts)
 * up to `capacity`; returns the number of rows the range produced, adds its counted columns to *n_cols.
 */
static int64_t count_sorted_entries(const entry_t* ents, int64_t n_ent, const uint8_t* const* ref, int32_t min_dp, int32_t min_cc,
                                    int64_t* out_keys, uint8_t* out_ref, uint32_t* out_counts, int64_t capacity, int64_t* n_cols, int32_t* cells)
{
    int64_t n_rows = 0, i = 0;
    while (i < n_ent) {
        int64_t j = i;
        while (j < n_ent && ents[j].key == ents[i].key) ++j;
        int32_t tid = (int32_t)(ents[i].key >> 32);
        int64_t pos = ents[i].key & 0xffffffffll;
        /* windows start at 1: 0-based position 0 is never visited (MakeWindows, :86) */
        if (pos >= 1) {
            ++*n_cols;
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
    return n_rows;
}

/* read admission: pysam pileup flag_filter + min_mapping_quality + ignore_orphans (BaseCellCounter.py:191),
   is_secondary/is_duplicate/is_supplementary (:249), CB present (:240-243), barcode belongs to this cell type's BAM
   (SplitBamCellTypes.py:83-90,110-113,173); malformed segments are never counted */
typedef struct {
    const int32_t* read_tid; const uint16_t* read_flag; const uint8_t* read_mapq; const int32_t* read_cb;
    const uint32_t* seg_read; const int32_t* seg_start; const int32_t* seg_len; const int64_t* seg_ev_off; const uint16_t* events;
    int32_t n_contigs; const int64_t* contig_len; const uint8_t* celltype_of; int32_t n_cb, ct, min_bq, min_mq; uint32_t flag_exclude; int32_t ignore_orphans;
} admit_t;

static int seg_admitted(const admit_t* a, int64_t s) {
    uint32_t r = a->seg_read[s];
    uint32_t flag = a->read_flag[r];
    if (flag & a->flag_exclude) return 0;
    if ((int)a->read_mapq[r] < a->min_mq) return 0;
    if (a->ignore_orphans && (flag & 1u) && !(flag & 2u)) return 0;
    int32_t cb = a->read_cb[r];
    if (cb < 0 || cb >= a->n_cb) return 0;
    if ((int)a->celltype_of[cb] != a->ct) return 0;
    int32_t tid = a->read_tid[r];
    if (tid < 0 || tid >= a->n_contigs) return 0;
    int32_t st = a->seg_start[s], ln = a->seg_len[s];
    if (st < 0 || ln <= 0 || (int64_t)st + ln > a->contig_len[tid]) return 0;
    return 1;
}

/* entries of segment s with position in [lo, hi): write to out (may be NULL = count only) */
static int64_t expand_segment(const admit_t* a, int64_t s, int64_t lo, int64_t hi, entry_t* out) {
    uint32_t r = a->seg_read[s];
    int32_t tid = a->read_tid[r], cb = a->read_cb[r];
    uint32_t flag = a->read_flag[r];
    int32_t st = a->seg_start[s], ln = a->seg_len[s];
    int64_t i0 = lo > st ? lo - st : 0, i1 = hi < (int64_t)st + ln ? hi - st : ln, w = 0;
    for (int64_t i = i0; i < i1; ++i) {
        uint16_t ev = a->events[a->seg_ev_off[s] + i];
        int q = ev & 0xff, sym = (ev >> 8) & 7;   /* LSG_EVENT: 0x0800 | class << 8 | qual, 0 = 'NA' */
        if (!(ev & 0x0800)) continue;      /* 'NA' symbols: not in BASE_COUNTS.keys() (:258) */
        if (q < a->min_bq) continue;       /* pileup_base_qual_skip: every accessor drops it */
        if (out) {
            entry_t* e = &out[w];
            e->key = ((int64_t)tid << 32) | (int64_t)(st + i);
            e->cb = cb; e->sym = (uint8_t)sym; e->qual = (uint8_t)q; e->rev = (uint8_t)((flag >> 4) & 1u); e->pad = 0;
        }
        ++w;
    }
    return w;
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
    admit_t a = { read_tid, read_flag, read_mapq, read_cb, seg_read, seg_start, seg_len, seg_ev_off, events,
                  n_contigs, contig_len, celltype_of, n_cb, ct, min_bq, min_mq, flag_exclude, ignore_orphans };
    /* pass 1: how many entries survive read admission + the base-quality gate; pass 2: expand them */
    int64_t n_ent = 0;
    for (int64_t s = 0; s < n_segs; ++s) if (seg_admitted(&a, s)) n_ent += expand_segment(&a, s, 0, INT64_MAX, NULL);
    entry_t* ents = (entry_t*)malloc(sizeof(entry_t) * (size_t)(n_ent > 0 ? n_ent : 1));
    int32_t* cells = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_ent > 0 ? n_ent : 1));
    if (!ents || !cells) { free(ents); free(cells); return -1; }
    int64_t w = 0;
    for (int64_t s = 0; s < n_segs; ++s) if (seg_admitted(&a, s)) w += expand_segment(&a, s, 0, INT64_MAX, ents + w);
    qsort(ents, (size_t)n_ent, sizeof(entry_t), cmp_entry);
    int64_t n_cols = 0;
    int64_t n_rows = count_sorted_entries(ents, n_ent, ref, min_dp, min_cc, out_keys, out_ref, out_counts, capacity, &n_cols, cells);
    free(cells); free(ents);
    if (n_columns) *n_columns = n_cols;
    return n_rows;
}


/* ------------------------------------------------------------------------------------------------
 * Region-parallel form of lso_count: the genome is cut into regions of `region_w` positions, every admitted
 * segment is listed in the regions it overlaps, and threads take regions off a shared counter; a region's
 * entries are expanded, sorted and counted by the SAME functions lso_count uses (columns are independent,
 * BaseCellCounter.py:200 does the same with its 50 kb windows and a process pool, :392-402).  Rows come back
 * in (tid, pos) order.  Used for the oracle hashes of the multi-million-read samples (tools/oracle_hashes.py)
 * and for bench.py's all-cores CPU baseline.
 */
#include <pthread.h>

typedef struct { int64_t n_rows, n_cols; int64_t* keys; uint8_t* ref; uint32_t* counts; } region_out_t;
typedef struct {
    const admit_t* a; const uint8_t* const* ref; int32_t min_dp, min_cc, region_w;
    int64_t n_regions; const int64_t* region_base;      /* first region of every contig */
    const int64_t* seg_off; const int64_t* seg_list;    /* segments per region */
    region_out_t* out; int64_t next; pthread_mutex_t mu; int failed;
    int64_t key_lo, key_hi;                             /* counted span [(tid << 32 | pos) lo, hi): lso_count_span_mt */
} mt_t;

static void* mt_worker(void* arg) {
    mt_t* m = (mt_t*)arg;
    entry_t* ents = NULL; int32_t* cells = NULL; int64_t cap = 0;
    for (;;) {
        pthread_mutex_lock(&m->mu);
        int64_t rg = m->next++;
        pthread_mutex_unlock(&m->mu);
        if (rg >= m->n_regions) break;
        int64_t s0 = m->seg_off[rg], s1 = m->seg_off[rg + 1];
        if (s0 == s1) continue;
        int32_t tid = 0;
        while (tid + 1 < m->a->n_contigs && m->region_base[tid + 1] <= rg) ++tid;
        int64_t lo = (rg - m->region_base[tid]) * (int64_t)m->region_w, hi = lo + m->region_w;
        {   /* columns outside the counted span belong to another shard (POS >= START and POS < END, BaseCellCounter.py:200) */
            const int64_t klo = ((int64_t)tid << 32) | lo, khi = ((int64_t)tid << 32) | hi;
            if (khi <= m->key_lo || klo >= m->key_hi) continue;
            if (klo < m->key_lo) lo = m->key_lo & 0xffffffffll;
            if (khi > m->key_hi) hi = m->key_hi & 0xffffffffll;
        }
        int64_t n_ent = 0;
        for (int64_t i = s0; i < s1; ++i) n_ent += expand_segment(m->a, m->seg_list[i], lo, hi, NULL);
        if (n_ent == 0) continue;
        if (n_ent > cap) {
            free(ents); free(cells); cap = n_ent + n_ent / 4;
            ents = (entry_t*)malloc(sizeof(entry_t) * (size_t)cap); cells = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
            if (!ents || !cells) { m->failed = 1; break; }
        }
        int64_t w = 0;
        for (int64_t i = s0; i < s1; ++i) w += expand_segment(m->a, m->seg_list[i], lo, hi, ents + w);
        qsort(ents, (size_t)n_ent, sizeof(entry_t), cmp_entry);
        region_out_t* o = &m->out[rg];
        o->keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)m->region_w);
        o->ref = (uint8_t*)malloc((size_t)m->region_w);
        o->counts = (uint32_t*)malloc(sizeof(uint32_t) * 42 * (size_t)m->region_w);
        if (!o->keys || !o->ref || !o->counts) { m->failed = 1; break; }
        o->n_rows = count_sorted_entries(ents, n_ent, m->ref, m->min_dp, m->min_cc, o->keys, o->ref, o->counts, m->region_w, &o->n_cols, cells);
    }
    free(ents); free(cells);
    return NULL;
}

/* the same, restricted to the columns with key (tid << 32 | pos) in [key_lo, key_hi): one shard of a job whose reads are
   generated shard by shard (tools/oracle_hashes.py streams the full-size workloads through it) */
int64_t lso_count_span_mt(int64_t n_reads, int64_t n_segs,
                     const int32_t* read_tid, const uint16_t* read_flag, const uint8_t* read_mapq, const int32_t* read_cb,
                     const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, const int64_t* seg_ev_off,
                     const uint16_t* events,
                     int32_t n_contigs, const int64_t* contig_len, const uint8_t* const* ref,
                     const uint8_t* celltype_of, int32_t n_cb, int32_t ct,
                     int32_t min_bq, int32_t min_mq, int32_t min_dp, int32_t min_cc, uint32_t flag_exclude, int32_t ignore_orphans,
                     int64_t* out_keys, uint8_t* out_ref, uint32_t* out_counts, int64_t capacity, int64_t* n_columns,
                     int32_t n_threads, int32_t region_w, int64_t key_lo, int64_t key_hi)
{
    (void)n_reads;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (region_w < 1) region_w = 512;
    admit_t a = { read_tid, read_flag, read_mapq, read_cb, seg_read, seg_start, seg_len, seg_ev_off, events,
                  n_contigs, contig_len, celltype_of, n_cb, ct, min_bq, min_mq, flag_exclude, ignore_orphans };
    int64_t* base = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_contigs + 1));
    if (!base) return -1;
    base[0] = 0;
    for (int32_t t = 0; t < n_contigs; ++t) base[t + 1] = base[t] + (contig_len[t] + region_w - 1) / region_w;
    const int64_t n_regions = base[n_contigs];
    int64_t* off = (int64_t*)calloc((size_t)n_regions + 2, sizeof(int64_t));
    if (!off) { free(base); return -1; }
    for (int pass = 0; pass < 2; ++pass) {
        int64_t* list = NULL;
        if (pass == 1) {
            int64_t run = 0;
            for (int64_t r = 0; r <= n_regions; ++r) { int64_t c = off[r]; off[r] = run; run += c; }
            list = (int64_t*)malloc(sizeof(int64_t) * (size_t)(run > 0 ? run : 1));
            int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_regions + 1));
            if (!list || !cur) { free(list); free(cur); free(off); free(base); return -1; }
            memcpy(cur, off, sizeof(int64_t) * (size_t)(n_regions + 1));
            for (int64_t s = 0; s < n_segs; ++s) {
                if (!seg_admitted(&a, s)) continue;
                int32_t tid = read_tid[seg_read[s]];
                int64_t r0 = base[tid] + seg_start[s] / region_w, r1 = base[tid] + ((int64_t)seg_start[s] + seg_len[s] - 1) / region_w;
                for (int64_t r = r0; r <= r1; ++r) list[cur[r]++] = s;
            }
            free(cur);
            region_out_t* out = (region_out_t*)calloc((size_t)(n_regions > 0 ? n_regions : 1), sizeof(region_out_t));
            if (!out) { free(list); free(off); free(base); return -1; }
            mt_t m; memset(&m, 0, sizeof(m));
            m.a = &a; m.ref = ref; m.min_dp = min_dp; m.min_cc = min_cc; m.region_w = region_w; m.n_regions = n_regions; m.region_base = base;
            m.seg_off = off; m.seg_list = list; m.out = out; m.next = 0; m.failed = 0; m.key_lo = key_lo; m.key_hi = key_hi;
            pthread_mutex_init(&m.mu, NULL);
            pthread_t th[256];
            int started = 0;
            for (int i = 0; i < n_threads; ++i) if (pthread_create(&th[i], NULL, mt_worker, &m) == 0) ++started; else break;
            if (started == 0) mt_worker(&m);
            for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
            pthread_mutex_destroy(&m.mu);
            int64_t n_rows = 0, n_cols = 0;
            for (int64_t r = 0; r < n_regions; ++r) {
                region_out_t* o = &out[r];
                n_cols += o->n_cols;
                for (int64_t i = 0; i < o->n_rows; ++i, ++n_rows) {
                    if (n_rows < capacity) {
                        out_keys[n_rows] = o->keys[i]; out_ref[n_rows] = o->ref[i];
                        memcpy(out_counts + n_rows * 42, o->counts + i * 42, 42 * sizeof(uint32_t));
                    }
                }
                free(o->keys); free(o->ref); free(o->counts);
            }
            free(out); free(list); free(off); free(base);
            if (m.failed) return -1;
            if (n_columns) *n_columns = n_cols;
            return n_rows;
        }
        for (int64_t s = 0; s < n_segs; ++s) {
            if (!seg_admitted(&a, s)) continue;
            int32_t tid = read_tid[seg_read[s]];
            int64_t r0 = base[tid] + seg_start[s] / region_w, r1 = base[tid] + ((int64_t)seg_start[s] + seg_len[s] - 1) / region_w;
            for (int64_t r = r0; r <= r1; ++r) ++off[r];
        }
    }
    return -1;
}

int64_t lso_count_mt(int64_t n_reads, int64_t n_segs,
                     const int32_t* read_tid, const uint16_t* read_flag, const uint8_t* read_mapq, const int32_t* read_cb,
                     const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, const int64_t* seg_ev_off,
                     const uint16_t* events,
                     int32_t n_contigs, const int64_t* contig_len, const uint8_t* const* ref,
                     const uint8_t* celltype_of, int32_t n_cb, int32_t ct,
                     int32_t min_bq, int32_t min_mq, int32_t min_dp, int32_t min_cc, uint32_t flag_exclude, int32_t ignore_orphans,
                     int64_t* out_keys, uint8_t* out_ref, uint32_t* out_counts, int64_t capacity, int64_t* n_columns,
                     int32_t n_threads, int32_t region_w)
{
    return lso_count_span_mt(n_reads, n_segs, read_tid, read_flag, read_mapq, read_cb, seg_read, seg_start, seg_len, seg_ev_off, events,
                             n_contigs, contig_len, ref, celltype_of, n_cb, ct, min_bq, min_mq, min_dp, min_cc, flag_exclude, ignore_orphans,
                             out_keys, out_ref, out_counts, capacity, n_columns, n_threads, region_w, 0, INT64_MAX);
}
