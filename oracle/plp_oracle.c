/*
 * plp_oracle.c — TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * BAM-level, column-major restatement of what the reference does to one cell type's BAM:
 *   SplitBamCellTypes.split_bam   /root/reference/workflow/scripts/PreProcessing/SplitBamCellTypes.py:65-124,173
 *   BaseCellCounter.run_interval  /root/reference/workflow/scripts/SNVCalling/BaseCellCounter.py:182-320
 * including the third-party pileup engine it calls (NOT in the reference tree; SURVEY.md §8a rows
 * a4-a6): htslib's bam_plp column iterator with its per-read CIGAR cursor (resolve_cigar2) and
 * pysam's PileupColumn accessors (base-quality skip, get_query_sequences(add_indels=True)).  The
 * algorithm is stated the way those libraries run it — one column at a time, a cursor (op index k,
 * reference x, query y) per buffered read — so that it is an independent check of the product's
 * read-major decoder (longsom_amd/csrc/hostio/bamio.cpp) and of the HIP kernels.
 *
 * It has its own minimal BGZF/BAM reader on purpose (sequential zlib inflate).
 *
 * Parity status: "parity unpinned" against the reference itself for this stage — pysam/htslib are
 * not installed here and the reference ships no test vectors; the pin is the hand-derived known
 * answers in tests/golden/kat_pileup.json and by the tables the reference's own BaseCellCounter.py wrote over the column-replay
 * stand-in (tests/golden/pileup.*, tests/test_pileup_refgolden.py).  max_depth (BaseCellCounter.py:191: 200000) is
 * htslib's bam_plp_push rule: a read that starts at the iterator's current column — i.e. any read but the first of its start
 * position — is dropped while the buffer (reads not yet freed + the spare tail node) exceeds max_depth.  The reference opens a fresh
 * pileup per 50 kb window (BaseCellCounter.py:185-191, windows [1, 50001), [50001, 100001), ... from MakeWindows :81-113) over the reads
 * the index fetch returns for it, so with a cap the sweep below is made per window: its pool = the reads overlapping the window, its
 * columns = the window's (:200); without a cap the windows are result-neutral and one sweep per file is made.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* htslib <= 1.10 (resolve_cigar2 before the 1.11 rewrite of its indel peek): the last column of a D operation that is followed by
   another D carries indel < 0 too, so pysam prints "*-<n>N.." there and EasyReadPileup (BaseCellCounter.py:167-170) calls it 'D'
   instead of 'O'.  0 (default) = htslib >= 1.11. */
static int g_legacy_del_merge = 0;
void plp_set_legacy_del_merge(int on) { g_legacy_del_merge = on ? 1 : 0; }


typedef struct {
    int32_t tid, pos, end;          /* end = first reference position after the alignment */
    uint32_t flag, mapq, n_cigar, l_seq;
    const uint8_t *cigar, *seq, *qual;
    int32_t cb;                     /* dense barcode id, -1 none / unknown */
    /* pileup cursor */
    int32_t k; int64_t x; uint32_t y;
} read_t;

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
#define COP(c) ((c) & 0xf)
#define CLN(c) ((c) >> 4)
static int ref_op(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
static int match_op(uint32_t op) { return op == 0 || op == 7 || op == 8; }

static uint8_t* inflate_all(const char* path, size_t* out_len) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* raw = (uint8_t*)malloc(n ? n : 1);
    if (fread(raw, 1, n, f) != n) { fclose(f); free(raw); return NULL; }
    fclose(f);
    size_t cap = n * 4 + 65536, len = 0, off = 0;
    uint8_t* out = (uint8_t*)malloc(cap);
    while (off + 18 <= n) {
        const uint8_t* h = raw + off;
        uint32_t xlen = rd16(h + 10), bsize = rd16(h + 16) + 1;      /* BC subfield first, as every BAM writer emits it */
        uint32_t usize = rd32(h + bsize - 4);
        if (len + usize > cap) { cap = (len + usize) * 2; out = (uint8_t*)realloc(out, cap); }
        if (usize) {
            z_stream zs; memset(&zs, 0, sizeof(zs));
            inflateInit2(&zs, -15);
            zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = bsize - xlen - 20; zs.next_out = out + len; zs.avail_out = usize;
            int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (rc != Z_STREAM_END) { free(raw); free(out); return NULL; }
        }
        len += usize; off += bsize;
    }
    free(raw);
    *out_len = len;
    return out;
}

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return x < y ? -1 : x > y; }
static int distinct(int32_t* v, int n) {
    if (!n) return 0;
    qsort(v, (size_t)n, sizeof(int32_t), cmp_i32);
    int d = 1;
    for (int i = 1; i < n; ++i) d += v[i] != v[i - 1];
    return d;
}

/* htslib's per-read cursor: position the read on column `pos` and report qpos / is_del / is_refskip / indel. */
static void resolve(read_t* r, int64_t pos, uint32_t* qpos, int* is_del, int* is_refskip, int* indel) {
    const uint8_t* cg = r->cigar;
    if (r->k < 0) {                                   /* first column of this read: find the first M/D/N/=/X */
        r->x = r->pos; r->y = 0;
        uint32_t k;
        for (k = 0; k < r->n_cigar; ++k) {
            uint32_t c = rd32(cg + 4 * k), op = COP(c);
            if (ref_op(op)) break;
            if (op == 1 || op == 4) r->y += CLN(c);
        }
        r->k = (int32_t)k;
    }
    {
        uint32_t c = rd32(cg + 4 * r->k), l = CLN(c);
        while (pos - r->x >= (int64_t)l) {            /* the column left this operation: move to the next M/D/N/=/X (several, when columns
                                                         nobody looks at were skipped: the windowed sweep below) */
            if (match_op(COP(c))) r->y += l;
            r->x += l;
            uint32_t k;
            for (k = (uint32_t)r->k + 1; k < r->n_cigar; ++k) {
                uint32_t c2 = rd32(cg + 4 * k), op2 = COP(c2);
                if (ref_op(op2)) break;
                if (op2 == 1 || op2 == 4) r->y += CLN(c2);
            }
            r->k = (int32_t)k;
            c = rd32(cg + 4 * r->k); l = CLN(c);
        }
    }
    uint32_t c = rd32(cg + 4 * r->k), op = COP(c), l = CLN(c);
    *is_del = *is_refskip = *indel = 0;
    if (r->x + (int64_t)l - 1 == pos && (uint32_t)r->k + 1 < r->n_cigar) {     /* last column of the op: peek */
        uint32_t c2 = rd32(cg + 4 * (r->k + 1)), op2 = COP(c2);
        if (op2 == 2 && g_legacy_del_merge) {         /* htslib <= 1.10: any operation's last column before a D, consecutive D's not summed */
            *indel = -(int)CLN(c2);
        } else if (op2 == 2 && op != 2) {             /* a deletion starts after this column (1D2D counts as 3D) */
            int d = (int)CLN(c2);
            for (uint32_t k = (uint32_t)r->k + 2; k < r->n_cigar; ++k) { uint32_t c3 = rd32(cg + 4 * k); if (COP(c3) == 2) d += (int)CLN(c3); else break; }
            *indel = -d;
        } else if (op2 == 1) {                        /* insertion (pads between insertions are skipped) */
            int d = (int)CLN(c2);
            for (uint32_t k = (uint32_t)r->k + 2; k < r->n_cigar; ++k) {
                uint32_t c3 = rd32(cg + 4 * k), o3 = COP(c3);
                if (o3 == 1) d += (int)CLN(c3); else if (o3 != 6) break;
            }
            *indel = d;
        } else if (op2 == 6 && (uint32_t)r->k + 2 < r->n_cigar) {              /* pad, then maybe an insertion */
            int d = 0;
            for (uint32_t k = (uint32_t)r->k + 2; k < r->n_cigar; ++k) {
                uint32_t c3 = rd32(cg + 4 * k), o3 = COP(c3);
                if (o3 == 1) d += (int)CLN(c3); else if (ref_op(o3)) break;
            }
            if (d > 0) *indel = d;
        }
    }
    if (match_op(op)) *qpos = r->y + (uint32_t)(pos - r->x);
    else { *is_del = 1; *qpos = r->y; *is_refskip = op == 3; }                 /* D or N: query index of the next base */
}

/* symbol classes A C T G I D N O = 0..7, NA = 15 */
static int easy_read_pileup(const read_t* r, uint32_t qpos, int is_del, int is_refskip, int indel) {
    /* pysam builds  <char>[+n<seq> | -n<N..>]  per entry; EasyReadPileup (BaseCellCounter.py:152-180) keys on
       len == 1 and the upper-cased char, or on x[1] when an indel suffix follows */
    if (indel < 0) return 5;
    if (indel > 0) return 4;
    if (is_del) return is_refskip ? 15 : 7;           /* '>' '<' -> NA ; '*' -> O */
    if (qpos >= r->l_seq) return 6;                   /* pysam prints 'N' beyond the stored sequence */
    uint32_t code = (r->seq[qpos >> 1] >> ((~qpos & 1u) << 2)) & 0xf;          /* "=ACMGRSVTWYHKDBN" */
    switch (code) { case 1: return 0; case 2: return 1; case 8: return 2; case 4: return 3; case 15: return 6; default: return 15; }
}

/*
 * barcodes: n_cb cleaned barcode strings joined by '\n' (dense id = index); celltype_of[id] cell type.
 * Returns rows of cell type `ct` (42 words each, as count_oracle.c), or < 0 on error.
 */
int64_t plp_count(const char* bam_path, const char* barcodes, int32_t n_cb, const uint8_t* celltype_of, int32_t ct,
                  int32_t n_contigs, const int64_t* contig_len, const uint8_t* const* ref,
                  int32_t min_bq, int32_t min_mq, int32_t min_dp, int32_t min_cc,
                  int64_t* out_keys, uint8_t* out_ref, uint32_t* out_counts, int64_t capacity, int32_t max_depth, int32_t window)
{
    size_t len = 0;
    uint8_t* d = inflate_all(bam_path, &len);
    if (!d || len < 12 || memcmp(d, "BAM\1", 4)) { free(d); return -1; }
    size_t p = 8 + rd32(d + 4);
    uint32_t n_ref = rd32(d + p); p += 4;
    for (uint32_t i = 0; i < n_ref; ++i) { p += 4 + rd32(d + p); p += 4; }
    /* barcode table */
    const char** bc = (const char**)malloc(sizeof(char*) * (size_t)(n_cb ? n_cb : 1));
    size_t* bl = (size_t*)malloc(sizeof(size_t) * (size_t)(n_cb ? n_cb : 1));
    { const char* s = barcodes; for (int i = 0; i < n_cb; ++i) { const char* e = strchr(s, '\n'); size_t l = e ? (size_t)(e - s) : strlen(s); bc[i] = s; bl[i] = l; s += l + (e ? 1 : 0); } }
    /* records that survive SplitBamCellTypes for this cell type, then the pileup engine's read filter */
    size_t cap = 1024, n = 0;
    read_t* rd = (read_t*)malloc(sizeof(read_t) * cap);
    while (p + 4 <= len) {
        uint32_t bs = rd32(d + p); const uint8_t* r = d + p + 4; p += 4 + bs;
        read_t t; memset(&t, 0, sizeof(t));
        t.tid = (int32_t)rd32(r); t.pos = (int32_t)rd32(r + 4);
        uint32_t l_name = r[8]; t.mapq = r[9]; t.n_cigar = rd16(r + 12); t.flag = rd16(r + 14); t.l_seq = rd32(r + 16);
        t.cigar = r + 32 + l_name; t.seq = t.cigar + 4 * t.n_cigar; t.qual = t.seq + (t.l_seq + 1) / 2;
        const uint8_t* aux = t.qual + t.l_seq; const uint8_t* end = r + bs;
        if (t.tid < 0) continue;
        /* read.opt("CB") */
        const char* cb = NULL; size_t cbl = 0;
        while (aux + 3 <= end) {
            char a0 = (char)aux[0], a1 = (char)aux[1], ty = (char)aux[2]; aux += 3; size_t sz;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1; else if (ty == 's' || ty == 'S') sz = 2; else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { sz = strlen((const char*)aux) + 1; if (a0 == 'C' && a1 == 'B' && ty == 'Z') { cb = (const char*)aux; cbl = sz - 1; } }
            else if (ty == 'B') { char st = (char)aux[0]; uint32_t cnt = rd32(aux + 1); sz = 5 + (size_t)cnt * ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4); }
            else break;
            aux += sz;
        }
        if (!cb) continue;                                            /* CB_not_found */
        size_t cl = 0; while (cl < cbl && cb[cl] != '-') ++cl;        /* barcode.split("-")[0] */
        t.cb = -1;
        for (int i = 0; i < n_cb; ++i) if (bl[i] == cl && !memcmp(bc[i], cb, cl)) t.cb = i;   /* last match wins */
        if (t.cb < 0 || celltype_of[t.cb] != ct) continue;            /* CB_not_matched / other cell type's BAM */
        if ((int)t.mapq < min_mq) continue;                           /* SplitBam MAPQ filter (min_MQ = min_mq in LongSom's rules) */
        /* pileup engine: flag_filter UNMAP|SECONDARY|QCFAIL|DUP, min_mapping_quality, ignore_orphans */
        if (t.flag & (0x4 | 0x100 | 0x200 | 0x400)) continue;
        if ((t.flag & 0x1) && !(t.flag & 0x2)) continue;
        if (t.n_cigar == 0) continue;
        int64_t e = t.pos;
        for (uint32_t k = 0; k < t.n_cigar; ++k) { uint32_t c = rd32(t.cigar + 4 * k); if (ref_op(COP(c))) e += CLN(c); }
        t.end = (int32_t)(e > t.pos ? e : t.pos + 1);
        t.k = -1;
        if (n == cap) { cap *= 2; rd = (read_t*)realloc(rd, sizeof(read_t) * cap); }
        rd[n++] = t;
    }
    /* column sweeps (input is coordinate sorted): one per pileup call of the reference */
    int64_t n_rows = 0;
    read_t** act = (read_t**)malloc(sizeof(read_t*) * (n ? n : 1));
    read_t** L = (read_t**)malloc(sizeof(read_t*) * (n ? n : 1));
    int32_t* cells = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    int* syms = (int*)malloc(sizeof(int) * (n ? n : 1));
    uint32_t* quals = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    const int per_window = max_depth > 0 && window > 0;
    size_t c0 = 0;                                                       /* first read of the current contig */
    for (int32_t wt = 0; wt < (per_window ? n_contigs : 1); ++wt) {
      while (per_window && c0 < n && rd[c0].tid < wt) ++c0;
      size_t c1 = c0; while (per_window && c1 < n && rd[c1].tid == wt) ++c1;
      const int64_t wlen = per_window ? contig_len[wt] : 0;
      for (int64_t ws = 1; per_window ? ws < wlen : ws == 1; ws += (per_window ? window : 1)) {
        const int64_t elo = per_window ? ws : 1, ehi = per_window ? ws + window : INT64_MAX;
        size_t nl = 0;
        if (per_window) { for (size_t i = c0; i < c1 && rd[i].pos < ehi; ++i) if (rd[i].end > elo) L[nl++] = &rd[i]; }      /* the index fetch: reads overlapping [ws, we) */
        else for (size_t i = 0; i < n; ++i) L[nl++] = &rd[i];
        if (!nl) continue;
        size_t head = 0, n_act = 0;
        int32_t cur_tid = -1; int64_t pos = 0;
        for (size_t i = 0; i < nl; ++i) L[i]->k = -1;                  /* fresh pileup: the reads' CIGAR cursors start over */
    while (head < nl || n_act) {
        if (!n_act) { cur_tid = L[head]->tid; pos = L[head]->pos; }
        {   /* bam_plp_push: the first read of a start position always enters; later ones only while mp->cnt (= buffered + 1) <= maxcnt.
               The buffer still holds the reads that ended on the previous column (freed by the sweep below). */
            int first_here = 1;
            while (head < nl && L[head]->tid == cur_tid && L[head]->pos <= pos) {
                if (max_depth > 0 && L[head]->pos == pos && !first_here && (int64_t)n_act + 1 > (int64_t)max_depth) { ++head; continue; }
                first_here = 0;
                act[n_act++] = L[head++];
            }
        }
        /* drop finished reads */
        size_t w = 0;
        for (size_t i = 0; i < n_act; ++i) if (act[i]->end > pos) act[w++] = act[i];
        n_act = w;
        if (!n_act) { if (head < nl && L[head]->tid == cur_tid && L[head]->pos > pos) pos = L[head]->pos; else if (head < nl) { cur_tid = -1; } continue; }
        /* Columns below the window are yielded by pysam and skipped by the script (:200): nothing of them is looked at but what the
           buffer holds when the next read is pushed, so the sweep jumps to the next read's start (or the window's) - the reads that end
           in between are freed before that push exactly as the skipped columns would have freed them.  Past the window nothing is left to do. */
        if (pos >= ehi) break;
        if (pos < elo) {
            int64_t nx = elo;
            if (head < nl && L[head]->tid == cur_tid && L[head]->pos < nx) nx = L[head]->pos;
            if (nx <= pos) nx = pos + 1;
            pos = nx;
            w = 0;
            for (size_t i = 0; i < n_act; ++i) if (act[i]->end > pos - 1) act[w++] = act[i];
            n_act = w;
            continue;
        }
        /* one pileup column */
        int m = 0;
        for (size_t i = 0; i < n_act; ++i) {
            uint32_t qpos; int is_del, is_refskip, indel;
            resolve(act[i], pos, &qpos, &is_del, &is_refskip, &indel);
            uint32_t q = qpos < act[i]->l_seq ? act[i]->qual[qpos] : 0;               /* pileup_base_qual_skip */
            if ((int)q < min_bq) continue;
            syms[m] = easy_read_pileup(act[i], qpos, is_del, is_refskip, indel); quals[m] = q; cells[m] = (int32_t)i; ++m;
        }
        /* windows start at 1 (MakeWindows): column 0 is never visited; DP gate :211; callable gate :220-222 */
        int64_t clen = (cur_tid < n_contigs) ? contig_len[cur_tid] : 0;
        uint8_t refb = (ref && cur_tid < n_contigs && ref[cur_tid] && pos < clen) ? ref[cur_tid][pos] : (uint8_t)'?';
        int non_na = 0; for (int i = 0; i < m; ++i) non_na += syms[i] != 15;
        if (pos >= elo && pos < ehi && pos < clen && m >= min_dp && refb != 'N' && non_na >= min_dp) {
            uint32_t bcn[8] = {0}, bq[8] = {0}, bcf[8] = {0}, bcr[8] = {0}, cc[8] = {0}, count = 0;
            for (int i = 0; i < m; ++i) {
                read_t* r = act[cells[i]];
                if (r->flag & (0x100 | 0x400 | 0x800)) continue;                     /* :249 */
                int s = syms[i]; if (s == 15) continue;                              /* :258 */
                ++count; ++bcn[s]; bq[s] += quals[i];
                if (r->flag & 0x10) ++bcr[s]; else ++bcf[s];
            }
            if ((int)count >= min_dp) {                                             /* :282 */
                int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m ? m : 1));
                int t_all = 0;
                for (int s = 0; s < 8; ++s) {
                    int t = 0;
                    for (int i = 0; i < m; ++i) { read_t* r = act[cells[i]]; if (!(r->flag & (0x100 | 0x400 | 0x800)) && syms[i] == s) tmp[t++] = r->cb; }
                    cc[s] = (uint32_t)distinct(tmp, t);
                }
                for (int i = 0; i < m; ++i) { read_t* r = act[cells[i]]; if (!(r->flag & (0x100 | 0x400 | 0x800)) && syms[i] != 15) tmp[t_all++] = r->cb; }
                uint32_t nc = (uint32_t)distinct(tmp, t_all);
                free(tmp);
                if ((int)nc >= min_cc) {                                            /* :294 */
                    if (n_rows < capacity) {
                        uint32_t* o = out_counts + n_rows * 42;
                        out_keys[n_rows] = ((int64_t)cur_tid << 32) | pos; out_ref[n_rows] = refb;
                        o[0] = count; o[1] = nc;
                        for (int s = 0; s < 8; ++s) { o[2 + s] = cc[s]; o[10 + s] = bcn[s]; o[18 + s] = bq[s]; o[26 + s] = bcf[s]; o[34 + s] = bcr[s]; }
                    }
                    ++n_rows;
                }
            }
        }
        ++pos;
    }
      }
    }
    free(L);
    free(act); free(cells); free(syms); free(quals); free(rd); free(bc); free(bl); free(d);
    return n_rows;
}
