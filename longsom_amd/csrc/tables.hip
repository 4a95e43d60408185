// The text of the reference's three big tables, printed on the device.
//
// Replaces the writers of
//   BaseCellCounter TSV rows     workflow/scripts/SNVCalling/BaseCellCounter.py:300-308
//   merged TSV rows              workflow/scripts/SNVCalling/MergeBaseCellCounts.py:48-84,116-204
//   step-1 TSV rows              workflow/scripts/SNVCalling/BaseCellCalling.step1.py:430-476
// (and the awk filter of BaseCellCalling.step2.py:23 for the rows step 2 keeps).  At C2 these are 16.6 GB of text for 24 M sites; on the
// host's cores (hostio/tsvwrite.cpp, the same rules, kept as the test reference and for the per-region pieces) printing them is what an
// end-to-end run waits for.  Here one thread prints one row, twice: a first pass with a sink that only counts gives every row its
// length, a scan gives it its place (the contigs in Python string order: a row's place is its genomic-order offset plus its contig's
// shift), a second pass with a sink that stores writes the bytes.  The host moves bytes: lsg_append_table streams the buffer through two
// pinned staging buffers into pwrite.
#include "lsg_ctx.h"
#include "call_rec.h"
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <hipcub/hipcub.hpp>

namespace lsg {
int run_export_rows(lsg_ctx* c, int ct, DevBuf& dk, DevBuf& dr, DevBuf& dc);

namespace {

__device__ __forceinline__ int n_digits(uint64_t v) {
    int n = 1;
    while (v >= 10000) { v /= 10000; n += 4; }
    return n + (v >= 1000 ? 3 : v >= 100 ? 2 : v >= 10 ? 1 : 0);
}

struct LenSink {
    uint32_t n = 0;
    __device__ __forceinline__ void ch(char) { ++n; }
    __device__ __forceinline__ void str(const char*, int k) { n += (uint32_t)k; }
    __device__ __forceinline__ void u64(uint64_t v) { n += (uint32_t)n_digits(v); }
};
struct PutSink {
    char* p;
    __device__ __forceinline__ void ch(char c) { *p++ = c; }
    __device__ __forceinline__ void str(const char* s, int k) { for (int i = 0; i < k; ++i) p[i] = s[i]; p += k; }
    __device__ __forceinline__ void u64(uint64_t v) {
        const int d = n_digits(v);
        char* e = p + d;
        p = e;
        do { const uint64_t q = v / 10; *--e = (char)('0' + (int)(v - q * 10)); v = q; } while (v);
    }
};
#define LIT(s, text) (s).str(text, (int)sizeof(text) - 1)

template <class S> __device__ __forceinline__ void put_i64(S& s, int64_t v) {
    if (v < 0) { s.ch('-'); s.u64((uint64_t)(-v)); } else s.u64((uint64_t)v);
}
// repr(k / 10000.0) for the integer k = round(p, 4) * 1e4: at least one decimal, trailing zeros cut (tsvwrite.cpp put_p4)
template <class S> __device__ __forceinline__ void put_p4(S& s, int64_t k) {
    if (k < 0) { s.ch('-'); k = -k; }
    s.u64((uint64_t)(k / 10000));
    s.ch('.');
    const int f = (int)(k % 10000);
    const char d[4] = {(char)('0' + f / 1000), (char)('0' + f / 100 % 10), (char)('0' + f / 10 % 10), (char)('0' + f % 10)};
    int n = 4;
    while (n > 1 && d[n - 1] == '0') --n;
    for (int i = 0; i < n; ++i) s.ch(d[i]);
}
// str(round(a / float(b), 4)): the double quotient, its EXACT binary value rounded half-even to 4 decimals (what glibc's "%.4f" prints
// in tsvwrite.cpp put_ratio).  x * 1e4 = hi + lo exactly (fma); the fraction of hi against 1/2, then lo, decide.
template <class S> __device__ __forceinline__ void put_ratio(S& s, int64_t a, int64_t b) {
    if (b == 0) { LIT(s, "nan"); return; }          // (not reachable: a considered cell type has DP >= min_cov and NC >= min_cells)
    const double x = (double)a / (double)b;
    const double hi = x * 10000.0, lo = fma(x, 10000.0, -hi);
    const double fl = floor(hi);
    int64_t k = (int64_t)fl;
    const double t = ((hi - fl) - 0.5) + lo;        // (exact wherever its sign is in doubt: hi - fl is exact, and within [1/4, 3/4] so is the - 1/2)
    if (t > 0.0 || (t == 0.0 && (k & 1))) ++k;
    put_p4(s, k);
}
// 'DP|NC|CC|BC|BQ|BCf|BCr' values of one LSG_ROW_WORDS row: six printed classes per vector
template <class S> __device__ __forceinline__ void put_row(S& s, const uint32_t* c) {
    s.u64(c[0]); s.ch('|'); s.u64(c[1]);
    for (int o = 2; o < 42; o += 8) {
        s.ch('|');
        for (int k = 0; k < 6; ++k) { if (k) s.ch(':'); s.u64(c[o + k]); }
    }
}

struct FmtArgs {
    int32_t kind, ct, n_ct, n_contigs;
    int64_t n;                                        // rows of the cell type (counts) or merged sites
    const int64_t* keys[LSG_MAX_CELLTYPES]; const uint8_t* refs[LSG_MAX_CELLTYPES]; const uint32_t* rows[LSG_MAX_CELLTYPES]; int64_t n_rows[LSG_MAX_CELLTYPES];
    const SiteRec* sites; const CandCt* cands;
    const uint8_t* const* ref_ptr; const int64_t* contig_len;
    const char* contig_txt; const uint32_t* contig_off;       // contig t's name = contig_txt[contig_off[t] .. contig_off[t + 1])
    const char* ct_txt; const uint32_t* ct_off;
    uint32_t* len; const uint64_t* off; const int64_t* shift; char* text;
    const int64_t* posset[3]; int64_t n_posset[3];    // K_STEP2: the resident position sets (RNA editing, PoN_SR, PoN_LR)
};
enum { K_COUNTS = 0, K_MERGED, K_STEP1, K_KEPT, K_STEP2 };

__device__ __forceinline__ int64_t row_of(const FmtArgs& a, int ct, int64_t key) {
    const int64_t* k = a.keys[ct];
    int64_t lo = 0, hi = a.n_rows[ct];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < a.n_rows[ct] && k[lo] == key) ? lo : -1;
}
template <class S> __device__ __forceinline__ void put_name(S& s, const char* txt, const uint32_t* off, int i) {
    s.str(txt + off[i], (int)(off[i + 1] - off[i]));
}
__device__ __forceinline__ const char* ct_filter_name(int f, int& n) {
    switch (f) {
        case LSG_CF_NONSIG: n = 15; return "Non-Significant";
        case LSG_CF_LOWSIG: n = 16; return "Low-Significance";
        case LSG_CF_MULTI: n = 13; return "Multi-allelic";
        case LSG_CF_LOW_CELLS: n = 9; return "Low_cells";
        case LSG_CF_LOW_READS: n = 9; return "Low_reads";
        case LSG_CF_PASS: n = 4; return "PASS";
        default: n = 0; return "";
    }
}
__device__ __forceinline__ const char* site_filter_name(int bit, int& n) {
    switch (bit) {
        case 0: n = 19; return "Multiple_cell_types";
        case 1: n = 13; return "Multi-allelic";
        case 2: n = 14; return "Min_cell_types";
        case 3: n = 15; return "Cell_type_noise";
        case 4: n = 10; return "Noisy_site";
        case 5: n = 11; return "LC_Upstream";
        default: n = 13; return "LC_Downstream";
    }
}

template <class S> __device__ void count_row(S& s, const FmtArgs& a, int64_t i) {
    const int64_t k = a.keys[a.ct][i];
    put_name(s, a.contig_txt, a.contig_off, (int)(k >> 32)); s.ch('\t');
    s.u64((uint64_t)(k & 0xFFFFFFFFll) + 1); s.ch('\t');
    s.ch((char)a.refs[a.ct][i]); s.ch('\t'); LIT(s, "DP|NC|CC|BC|BQ|BCf|BCr"); s.ch('\t');
    put_row(s, a.rows[a.ct] + i * LSG_ROW_WORDS); s.ch('\n');
}

template <class S> __device__ void merged_row(S& s, const FmtArgs& a, int64_t i) {
    const int64_t k = a.sites[i].key;
    int64_t r[LSG_MAX_CELLTYPES];
    for (int c = 0; c < a.n_ct; ++c) r[c] = row_of(a, c, k);
    put_name(s, a.contig_txt, a.contig_off, (int)(k >> 32)); s.ch('\t');
    s.u64((uint64_t)(k & 0xFFFFFFFFll) + 1); s.ch('\t'); s.u64((uint64_t)(k & 0xFFFFFFFFll) + 1); s.ch('\t');
    // sort_set (MergeBaseCellCounts.py:48-57): distinct REFs by decreasing count, first-seen order on ties
    char seen[LSG_MAX_CELLTYPES]; int cnt[LSG_MAX_CELLTYPES]; int ns = 0;
    for (int c = 0; c < a.n_ct; ++c) {
        if (r[c] < 0) continue;
        const char b = (char)a.refs[c][r[c]];
        int q = 0; while (q < ns && seen[q] != b) ++q;
        if (q == ns) { seen[ns] = b; cnt[ns++] = 1; } else ++cnt[q];
    }
    int idx[LSG_MAX_CELLTYPES] = {0, 1, 2, 3};
    for (int x = 1; x < ns; ++x) { const int v = idx[x]; int y = x; while (y > 0 && cnt[idx[y - 1]] < cnt[v]) { idx[y] = idx[y - 1]; --y; } idx[y] = v; }
    for (int q = 0; q < ns; ++q) { if (q) s.ch('|'); s.ch(seen[idx[q]]); }
    s.ch('\t'); LIT(s, "DP|NC|CC|BC|BQ|BCf|BCr");
    for (int c = 0; c < a.n_ct; ++c) {
        s.ch('\t');
        if (r[c] < 0) LIT(s, "NA"); else put_row(s, a.rows[c] + r[c] * LSG_ROW_WORDS);
    }
    s.ch('\n');
}

__device__ __forceinline__ bool in_posset(const FmtArgs& a, int kind, int64_t key) {
    const int64_t* k = a.posset[kind];
    int64_t lo = 0, hi = a.n_posset[kind];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; }
    return lo < a.n_posset[kind] && k[lo] == key;
}

// one row of the step-1 table (tsvwrite.cpp lsio_write_step1_rows, from the compact records instead of their lsg_call expansion).
// STEP2: the row as BaseCellCalling.step2.py leaves it when it has no gnomAD source and no distance filter (calling._step2_scanned):
// FILTER tagged by the position sets (GetExtraFilters, step2.py:142-158: "RNA_editing_db", "PoN_SR", "PoN_LR"; a tag replaces a bare
// "PASS"), every field but the first that is exactly "NA" empty (what the pandas round trip of step2.py does to them).
template <bool STEP2, class S> __device__ void step1_row(S& s, const FmtArgs& a, int64_t i) {
    const SiteRec c = a.sites[i];
    const int64_t k = c.key;
    const int tid = (int)(k >> 32);
    const int64_t pos = k & 0xFFFFFFFFll;
    const uint32_t sf = c.site_filter;
    const bool cand = (sf & (uint32_t)LSG_SF_CANDIDATE) != 0;
    int64_t r[LSG_MAX_CELLTYPES];
    for (int ct = 0; ct < a.n_ct; ++ct) r[ct] = row_of(a, ct, k);
    put_name(s, a.contig_txt, a.contig_off, tid); s.ch('\t');
    s.u64((uint64_t)pos + 1); s.ch('\t'); s.u64((uint64_t)pos + 1); s.ch('\t');
    s.ch((char)c.ref); s.ch('\t');
    auto context = [&]() {                             // Up_context \t Down_context ("." "." when POS < 6, step1.py:98-105)
        if (pos >= 5) {
            const uint8_t* ref = a.ref_ptr[tid];
            for (int q = 0; q < 5; ++q) s.ch((char)ref[pos - 5 + q]);
            s.ch('\t');
            if (STEP2 && pos + 3 == a.contig_len[tid] && ref[pos + 1] == 'N' && ref[pos + 2] == 'A') return;      // (a two-base context that reads "NA")
            for (int q = 0; q < 5 && pos + 1 + q < a.contig_len[tid]; ++q) s.ch((char)ref[pos + 1 + q]);
        } else LIT(s, ".\t.");
    };
    auto rest = [&](int q) {                           // Rest_BC / Rest_CC
        const int64_t s_alt = q ? c.sum_alts_cc : c.sum_alts_bc, s_tot = q ? c.sum_nc : c.sum_dp, pk = q ? c.noise_p_cc : c.noise_p_bc;
        put_i64(s, s_alt); s.ch(';'); put_i64(s, s_tot); s.ch(';');
        if (c.sum_alts_bc == 0) s.ch('1'); else if (pk == -2) LIT(s, "nan"); else put_p4(s, pk);
    };
    if (cand) {
        const CandCt* d = a.cands + (uint64_t)c.cand * a.n_ct;
        auto n_alt = [&](int ct) { const int na = d[ct].n_alt; return na < LSG_CALL_MAX_ALT ? na : LSG_CALL_MAX_ALT; };
        auto column = [&](auto&& per_ct) {             // one value group per cell type with candidates, ',' between them
            bool first = true;
            for (int ct = 0; ct < a.n_ct; ++ct) {
                if (!((c.has_cand >> ct) & 1)) continue;
                if (!first) s.ch(',');
                first = false;
                per_ct(ct);
            }
        };
        auto per_alt = [&](int ct, auto&& one) { const int na = n_alt(ct); for (int q = 0; q < na; ++q) { if (q) s.ch('|'); one(q); } };
        auto filters = [&]() { column([&](int ct) { int n; const char* f = ct_filter_name(d[ct].ct_filter, n); s.str(f, n); }); };
        bool any_pass = false; int n_alt_str = 0;
        for (int ct = 0; ct < a.n_ct; ++ct) {
            if (!((c.has_cand >> ct) & 1)) continue;
            any_pass |= d[ct].ct_filter == LSG_CF_PASS;
            bool dup = false;                          // distinct ALT strings among the cell types (step1.py: len(set(ALTs)))
            for (int o = 0; o < ct; ++o) {
                if (!((c.has_cand >> o) & 1) || n_alt(o) != n_alt(ct)) continue;
                bool same = true;
                for (int q = 0; q < n_alt(ct); ++q) same &= (d[o].alt[q] & 3) == (d[ct].alt[q] & 3);
                dup |= same;
            }
            n_alt_str += !dup;
        }
        column([&](int ct) { per_alt(ct, [&](int q) { s.ch("ACTG"[d[ct].alt[q] & 3]); }); });
        s.ch('\t');
        bool any_site = false;
        for (int b = 0; b < 7; ++b) if (sf & (1u << b)) { if (any_site) s.ch(','); any_site = true; int n; const char* f = site_filter_name(b, n); s.str(f, n); }
        bool bare_pass = false;
        if (!any_site) { if (any_pass) bare_pass = true; else filters(); }
        if (STEP2) {
            const int64_t k1 = ((int64_t)tid << 32) | (pos + 1);
            const bool tag[3] = {in_posset(a, 0, k1), in_posset(a, 1, k1), in_posset(a, 2, k1)};
            for (int q = 0; q < 3; ++q) {
                if (!tag[q]) continue;
                if (bare_pass) bare_pass = false; else s.ch(',');
                if (q == 0) LIT(s, "RNA_editing_db"); else if (q == 1) LIT(s, "PoN_SR"); else LIT(s, "PoN_LR");
            }
        }
        if (bare_pass) LIT(s, "PASS");
        s.ch('\t');
        {
            int only = -1, n_with = 0;
            for (int ct = 0; ct < a.n_ct; ++ct) if ((c.has_cand >> ct) & 1) { only = ct; ++n_with; }
            const bool reads_na = STEP2 && n_with == 1 && a.ct_off[only + 1] - a.ct_off[only] == 2 && a.ct_txt[a.ct_off[only]] == 'N' && a.ct_txt[a.ct_off[only] + 1] == 'A';
            if (!reads_na) column([&](int ct) { put_name(s, a.ct_txt, a.ct_off, ct); });
        }
        s.ch('\t'); context(); s.ch('\t');
        s.u64((uint64_t)n_alt_str);
        s.ch('\t'); column([&](int ct) { s.u64(a.rows[ct][r[ct] * LSG_ROW_WORDS]); });
        s.ch('\t'); column([&](int ct) { s.u64(a.rows[ct][r[ct] * LSG_ROW_WORDS + 1]); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { s.u64(d[ct].alt_bc[q]); }); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { s.u64(d[ct].alt_cc[q]); }); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { put_ratio(s, d[ct].alt_bc[q], a.rows[ct][r[ct] * LSG_ROW_WORDS]); }); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { put_ratio(s, d[ct].alt_cc[q], a.rows[ct][r[ct] * LSG_ROW_WORDS + 1]); }); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { put_p4(s, d[ct].p_bc[q]); }); });
        s.ch('\t'); column([&](int ct) { per_alt(ct, [&](int q) { put_p4(s, d[ct].p_cc[q]); }); });
        s.ch('\t'); put_i64(s, c.cell_types_min); s.ch('\t'); put_i64(s, c.cell_types_min);
        s.ch('\t'); rest(0); s.ch('\t'); rest(1); LIT(s, "\t.\t"); filters();
    } else {
        LIT(s, ".\t");
        if (sf & 16u) LIT(s, "Noisy_site"); else s.ch('.');
        LIT(s, "\t.\t"); context();
        LIT(s, "\t.\t.\t.\t.\t.\t.\t.\t.\t.\t"); put_i64(s, c.cell_types_min); s.ch('\t'); put_i64(s, c.cell_types_min);
        s.ch('\t'); rest(0); s.ch('\t'); rest(1); LIT(s, "\t.\t.");
    }
    s.ch('\t'); LIT(s, "DP|NC|CC|BC|BQ|BCf|BCr");
    for (int ct = 0; ct < a.n_ct; ++ct) {
        s.ch('\t');
        if (r[ct] < 0) { if (!STEP2) LIT(s, "NA"); } else put_row(s, a.rows[ct] + r[ct] * LSG_ROW_WORDS);
    }
    s.ch('\n');
}

template <class S> __device__ __forceinline__ void any_row(S& s, const FmtArgs& a, int64_t i) {
    switch (a.kind) {
        case K_COUNTS: count_row(s, a, i); break;
        case K_MERGED: merged_row(s, a, i); break;
        case K_STEP1: step1_row<false>(s, a, i); break;
        case K_KEPT: if (a.sites[i].site_filter & (uint32_t)LSG_SF_CANDIDATE) step1_row<false>(s, a, i); break;      // (a candidate's ALT and FILTER are never ".")
        default: if (a.sites[i].site_filter & (uint32_t)LSG_SF_CANDIDATE) step1_row<true>(s, a, i);
    }
}
__device__ __forceinline__ int64_t key_at(const FmtArgs& a, int64_t i) { return a.kind == K_COUNTS ? a.keys[a.ct][i] : a.sites[i].key; }

__global__ __launch_bounds__(256) void k_row_len(FmtArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.n) return;
    LenSink s;
    if (i < a.n) any_row(s, a, i);
    a.len[i] = s.n;
}
__global__ __launch_bounds__(256) void k_row_put(FmtArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    PutSink s{a.text + (int64_t)a.off[i] + a.shift[(int)(key_at(a, i) >> 32)]};
    any_row(s, a, i);
}
// lo[t] = first row of contig t (t = n_contigs: the row count); then, contig after contig in output order, shift[t] = where the contig's
// text starts in the output minus where it starts in genomic order
__global__ void k_contig_lo(FmtArgs a, int64_t* lo) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > a.n_contigs) return;
    const int64_t want = (int64_t)t << 32;
    int64_t l = 0, h = a.n;
    while (l < h) { const int64_t mid = (l + h) >> 1; if (key_at(a, mid) < want) l = mid + 1; else h = mid; }
    lo[t] = l;
}
__global__ void k_contig_shift(const int64_t* lo, const uint64_t* off, const int32_t* order, int n_contigs, int64_t* shift) {
    if (threadIdx.x || blockIdx.x) return;
    int64_t base = 0;
    for (int j = 0; j < n_contigs; ++j) {
        const int t = order[j];
        const int64_t b = (int64_t)off[lo[t]], e = (int64_t)off[lo[t + 1]];
        shift[t] = base - b;
        base += e - b;
    }
}

// ---- what step 3 wants to know about the step-2 table, read off its text where it lies ---------------------------------------------
// (calling.step3 parses only the rows that survive its FILTER patterns, given the kinds of cell every column holds over ALL rows: the
// host finds both with passes over the table's gigabytes, hostio/tsvstep3.cpp lsio_step3_column_kinds and tsvscan.cpp; these are the
// same rules over the same bytes, one thread per row.)
__device__ __forceinline__ bool text_eq(const char* f, int n, const char* lit, int m) {
    if (n != m) return false;
    for (int i = 0; i < n; ++i) if (f[i] != lit[i]) return false;
    return true;
}
#define TEXT_IS(f, n, lit) text_eq(f, n, lit, (int)sizeof(lit) - 1)
// the strings pandas.read_csv takes for a missing value (tsvstep3.cpp is_na)
__device__ bool text_is_na(const char* f, int n) {
    if (n == 0) return true;
    if (n > 8) return false;
    return TEXT_IS(f, n, "#N/A") || TEXT_IS(f, n, "#N/A N/A") || TEXT_IS(f, n, "#NA") || TEXT_IS(f, n, "-1.#IND") || TEXT_IS(f, n, "-1.#QNAN") || TEXT_IS(f, n, "-NaN") ||
           TEXT_IS(f, n, "-nan") || TEXT_IS(f, n, "1.#IND") || TEXT_IS(f, n, "1.#QNAN") || TEXT_IS(f, n, "<NA>") || TEXT_IS(f, n, "N/A") || TEXT_IS(f, n, "NA") ||
           TEXT_IS(f, n, "NULL") || TEXT_IS(f, n, "NaN") || TEXT_IS(f, n, "None") || TEXT_IS(f, n, "n/a") || TEXT_IS(f, n, "nan") || TEXT_IS(f, n, "null");
}
__device__ __forceinline__ bool is_dig(char c) { return c >= '0' && c <= '9'; }
__device__ __forceinline__ bool is_xdig(char c) { return is_dig(c) || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }
__device__ __forceinline__ char lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }
__device__ bool all_digits(const char* s, int n) { if (n <= 0) return false; for (int i = 0; i < n; ++i) if (!is_dig(s[i])) return false; return true; }
__device__ bool ci_prefix(const char* f, int n, const char* lit, int m) { if (n < m) return false; for (int i = 0; i < m; ++i) if (lower(f[i]) != lit[i]) return false; return true; }
// does strtod (C locale) take the whole field, trailing blanks aside?  (tsvstep3.cpp classify: a number pandas would read and print differently)
__device__ bool strtod_takes_all(const char* f, int n) {
    int i = 0;
    while (i < n && (f[i] == ' ' || (f[i] >= '\t' && f[i] <= '\r'))) ++i;
    if (i < n && (f[i] == '+' || f[i] == '-')) ++i;
    int end;
    if (ci_prefix(f + i, n - i, "inf", 3)) { end = i + 3; if (ci_prefix(f + end, n - end, "inity", 5)) end += 5; }
    else if (ci_prefix(f + i, n - i, "nan", 3)) {
        end = i + 3;
        if (end < n && f[end] == '(') {
            int k = end + 1;
            while (k < n && (is_dig(f[k]) || (lower(f[k]) >= 'a' && lower(f[k]) <= 'z') || f[k] == '_')) ++k;
            if (k < n && f[k] == ')') end = k + 1;
        }
    } else if (i + 1 < n && f[i] == '0' && (f[i + 1] == 'x' || f[i + 1] == 'X')) {
        int j = i + 2, nd = 0;
        while (j < n && is_xdig(f[j])) { ++j; ++nd; }
        if (j < n && f[j] == '.') { int j2 = j + 1, nf = 0; while (j2 < n && is_xdig(f[j2])) { ++j2; ++nf; } if (nd + nf > 0) { j = j2; nd += nf; } }
        if (nd == 0) end = i + 1;                       // only the "0" of "0x" is a number
        else {
            if (j < n && (f[j] == 'p' || f[j] == 'P')) { int k = j + 1; if (k < n && (f[k] == '+' || f[k] == '-')) ++k; if (k < n && is_dig(f[k])) { while (k < n && is_dig(f[k])) ++k; j = k; } }
            end = j;
        }
    } else {
        int j = i, nd = 0;
        while (j < n && is_dig(f[j])) { ++j; ++nd; }
        if (j < n && f[j] == '.') { int j2 = j + 1, nf = 0; while (j2 < n && is_dig(f[j2])) { ++j2; ++nf; } if (nd + nf > 0) { j = j2; nd += nf; } }
        if (nd == 0) j = 0;                             // no number: strtod hands the start of the field back ...
        else if (j < n && (f[j] == 'e' || f[j] == 'E')) { int k = j + 1; if (k < n && (f[k] == '+' || f[k] == '-')) ++k; if (k < n && is_dig(f[k])) { while (k < n && is_dig(f[k])) ++k; j = k; } }
        end = j;
    }
    while (end < n && f[end] == ' ') ++end;             // (... from where the host's check walks over blanks too: a field of blanks alone counts as taken)
    return end == n && end != 0;
}
enum { KIND_NA = 1, KIND_INT = 2, KIND_FLOAT = 4, KIND_ODD = 8, KIND_OTHER = 16 };
// the kind of one cell (tsvstep3.cpp classify, rule for rule)
__device__ int text_kind(const char* f, int n) {
    if (text_is_na(f, n)) return KIND_NA;
    const char* s = f; int m = n; bool neg = false;
    if (m && s[0] == '-') { neg = true; ++s; --m; }
    if (all_digits(s, m)) return ((m > 1 && s[0] == '0') || (neg && m == 1 && s[0] == '0') || m > 18) ? KIND_ODD : KIND_INT;
    int dot = -1;
    for (int i = 0; i < m; ++i) if (s[i] == '.') { dot = i; break; }
    if (dot >= 0 && all_digits(s, dot) && all_digits(s + dot + 1, m - dot - 1)) {
        const char* ip = s; const int ni = dot; const char* fp = s + dot + 1; const int nf = m - dot - 1;
        if (ni > 1 && ip[0] == '0') return KIND_ODD;
        if (nf > 1 && fp[nf - 1] == '0') return KIND_ODD;
        int lead = 0;
        while (lead < ni + nf && (lead < ni ? ip[lead] : fp[lead - ni]) == '0') ++lead;
        const int sig = ni + nf - lead;
        if (sig > 15 || ni > 15) return KIND_ODD;
        if (sig == 0) return (nf == 1 && ni == 1) ? KIND_FLOAT : KIND_ODD;      // "0.0" / "-0.0"
        if (ni == 1 && ip[0] == '0') { int z = 0; while (z < nf && fp[z] == '0') ++z; if (z >= 4) return KIND_ODD; }
        return KIND_FLOAT;
    }
    if (TEXT_IS(s, m, "inf")) return KIND_FLOAT;
    const char c0 = f[0];
    if (is_dig(c0) || c0 == '+' || c0 == '-' || c0 == '.' || c0 == ' ' || c0 == 'i' || c0 == 'I' || c0 == 'n' || c0 == 'N') {
        if (n >= 72) return KIND_OTHER;
        if (strtod_takes_all(f, n)) return KIND_ODD;
    }
    if (TEXT_IS(f, n, "True") || TEXT_IS(f, n, "False") || TEXT_IS(f, n, "TRUE") || TEXT_IS(f, n, "FALSE") || TEXT_IS(f, n, "true") || TEXT_IS(f, n, "false")) return KIND_ODD;
    return KIND_OTHER;
}

struct TextRows {                                      // the rows of a formatted table: row i = text[off[i] + shift[tid of row i] ..) of len[i] bytes (0: no row)
    const char* text; const uint32_t* len; const uint64_t* off; const int64_t* shift; const SiteRec* sites; int64_t n;
};
__device__ __forceinline__ const char* row_text(const TextRows& t, int64_t i) { return t.text + (int64_t)t.off[i] + t.shift[(int)(t.sites[i].key >> 32)]; }

// per column, the OR of the kinds of its cells over every row (a '#' cuts the row as pandas' comment="#" does; columns past the row's
// last field hold a missing value)
__global__ __launch_bounds__(256) void k_text_kinds(TextRows t, int n_cols, uint32_t* kinds) {
    __shared__ uint32_t sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.n && t.len[i] > 0) {
        const char* r = row_text(t, i);
        int e = (int)t.len[i] - 1;                      // (without the newline)
        for (int q = 0; q < e; ++q) if (r[q] == '#') { e = q; break; }
        int f0 = 0;
        for (int c = 0; c < n_cols && r[0] != '#'; ++c) {      // (a line that starts with '#' is a comment)
            int k;
            if (f0 > e) k = KIND_NA;
            else {
                int f1 = f0;
                while (f1 < e && r[f1] != '\t') ++f1;
                k = text_kind(r + f0, f1 - f0);
                f0 = f1 + 1;
            }
            if ((sh[c] & (uint32_t)k) == 0) atomicOr(&sh[c], (uint32_t)k);
        }
    }
    __syncthreads();
    if (threadIdx.x < n_cols && sh[threadIdx.x]) atomicOr(&kinds[threadIdx.x], sh[threadIdx.x]);
}

__device__ bool text_contains(const char* f, int n, const char* lit, int m) {
    for (int i = 0; i + m <= n; ++i) { int j = 0; while (j < m && f[i + j] == lit[j]) ++j; if (j == m) return true; }
    return false;
}
#define TEXT_HAS(f, n, lit) text_contains(f, n, lit, (int)sizeof(lit) - 1)
// the rows step 3 can keep (calling._step3_survivors): FILTER free of the patterns of BaseCellCalling.step3.py:49-52 (chrM rows:
// "Min|LR|gnomAD|LC|RNA") / :60-84 (the others), Cell_types (field 6) not "Non-Cancer"; a row of fewer than 7 fields is not kept
__global__ __launch_bounds__(256) void k_text_survivors(TextRows t, uint32_t* len2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > t.n) return;
    uint32_t keep = 0;
    if (i < t.n && t.len[i] > 0 && row_text(t, i)[0] != '#') {
        const char* r = row_text(t, i);
        const int e = (int)t.len[i] - 1;
        int fo[8], fl[8], nf = 0, f0 = 0;
        while (nf < 8 && f0 <= e) { int f1 = f0; while (f1 < e && r[f1] != '\t') ++f1; fo[nf] = f0; fl[nf] = f1 - f0; ++nf; f0 = f1 + 1; }
        if (nf > 6) {
            const char* F = r + fo[5]; const int n = fl[5];
            const bool is_m = TEXT_IS(r + fo[0], fl[0], "chrM");
            const bool dead = is_m ? (TEXT_HAS(F, n, "Min") || TEXT_HAS(F, n, "LR") || TEXT_HAS(F, n, "gnomAD") || TEXT_HAS(F, n, "LC") || TEXT_HAS(F, n, "RNA"))
                                   : (TEXT_HAS(F, n, "Min_cell_types") || TEXT_HAS(F, n, "Noisy_site") || TEXT_HAS(F, n, "LC_Upstream") || TEXT_HAS(F, n, "LC_Downstream") ||
                                      TEXT_HAS(F, n, "RNA_editing_db") || TEXT_HAS(F, n, "PoN") || TEXT_HAS(F, n, "Cell_type_noise") || TEXT_HAS(F, n, "gnomAD"));
            if (!dead && !TEXT_IS(r + fo[6], fl[6], "Non-Cancer")) keep = t.len[i];
        }
    }
    len2[i] = keep;
}
__global__ __launch_bounds__(256) void k_text_copy_rows(TextRows t, const uint32_t* len2, const uint64_t* off2, const int64_t* shift2, char* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.n || len2[i] == 0) return;
    const char* r = row_text(t, i);
    char* w = dst + (int64_t)off2[i] + shift2[(int)(t.sites[i].key >> 32)];
    for (uint32_t q = 0; q < len2[i]; ++q) w[q] = r[q];
}

struct Widen { __host__ __device__ __forceinline__ uint64_t operator()(const uint32_t& v) const { return (uint64_t)v; } };

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

std::vector<std::string> split_names(const char* joined, int n) {
    std::vector<std::string> out;
    const char* p = joined;
    for (int i = 0; i < n; ++i) {
        const char* e = strchr(p, '\n');
        if (!e) e = p + strlen(p);
        out.emplace_back(p, (size_t)(e - p));
        p = *e ? e + 1 : e;
    }
    return out;
}

} // namespace

int run_set_table_names(lsg_ctx* c, int32_t n_contigs, const char* contig_names, int32_t n_ct, const char* ct_names) {
    if (n_contigs != c->n_contigs) { set_error("lsg_set_table_names: %d contig names for %d contigs (lsg_set_contigs)", n_contigs, c->n_contigs); return -2; }
    if (n_ct < 1 || n_ct > LSG_MAX_CELLTYPES) { set_error("lsg_set_table_names: n_celltypes %d not in [1,%d]", n_ct, LSG_MAX_CELLTYPES); return -2; }
    const auto cn = split_names(contig_names, n_contigs), tn = split_names(ct_names, n_ct);
    // one buffer: contig offsets | cell-type offsets | contigs in Python string order | the names' bytes
    std::vector<uint32_t> coff(n_contigs + 1, 0), toff(n_ct + 1, 0);
    std::string ctxt, ttxt;
    for (int i = 0; i < n_contigs; ++i) { ctxt += cn[(size_t)i]; coff[(size_t)i + 1] = (uint32_t)ctxt.size(); }
    for (int i = 0; i < n_ct; ++i) { ttxt += tn[(size_t)i]; toff[(size_t)i + 1] = (uint32_t)ttxt.size(); }
    std::vector<int32_t> order(n_contigs);
    for (int i = 0; i < n_contigs; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cn[(size_t)x] < cn[(size_t)y]; });
    const size_t at_ct_off = (size_t)(n_contigs + 1) * 4, at_order = at_ct_off + (size_t)(n_ct + 1) * 4, at_ctxt = at_order + (size_t)n_contigs * 4,
                 at_ttxt = at_ctxt + ctxt.size(), total = at_ttxt + ttxt.size();
    std::vector<char> host(total + 8, 0);
    memcpy(host.data(), coff.data(), coff.size() * 4);
    memcpy(host.data() + at_ct_off, toff.data(), toff.size() * 4);
    if (n_contigs) memcpy(host.data() + at_order, order.data(), order.size() * 4);
    memcpy(host.data() + at_ctxt, ctxt.data(), ctxt.size());
    memcpy(host.data() + at_ttxt, ttxt.data(), ttxt.size());
    if (c->tab_names.reserve(host.size())) return -1;
    LSG_HIP(hipMemcpyAsync(c->tab_names.p, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    c->tab_n_contigs = n_contigs; c->tab_n_ct = n_ct;
    c->tab_ct_off_at = (uint32_t)at_ct_off; c->tab_order_at = (uint32_t)at_order; c->tab_contig_txt_at = (uint32_t)at_ctxt; c->tab_ct_txt_at = (uint32_t)at_ttxt;
    return 0;
}

int run_format_table(lsg_ctx* c, int32_t table, int64_t* n_bytes) {
    if (n_bytes) *n_bytes = 0;
    if (table < 0 || table >= LSG_TABLE_SLOTS) { set_error("lsg_format_table: no table %d", table); return -2; }
    if (!c->counted) { set_error("lsg_format_table: no count rows (lsg_pileup_count or lsg_load_counts first)"); return -2; }
    if (c->tab_n_contigs != c->n_contigs || c->tab_n_ct < c->n_ct) { set_error("lsg_format_table: lsg_set_table_names first (names of %d contigs and %d cell types)", c->n_contigs, c->n_ct); return -2; }
    if (table == LSG_TABLE_STEP3_ROWS) { set_error("lsg_format_table: table %d is made by lsg_step2_summary", table); return -2; }
    const bool counts = table < LSG_TABLE_MERGED;
    if (counts && table >= c->n_ct) { set_error("lsg_format_table: no cell type %d", table); return -2; }
    if (!counts && !c->called) { set_error("lsg_format_table: the merged and step-1 tables need lsg_call_step1 (its merged site list)"); return -2; }
    for (int t = 0; t < c->n_contigs; ++t)          // (REF of the count rows, the step-1 rows' contexts)
        if (!c->ref_ptr[t]) { set_error("lsg_format_table: reference of contig %d not loaded", t); return -2; }
    hipStream_t st = c->stream;
    // flat copies of the rows the table prints (kept until the next count)
    for (int ct = counts ? table : 0; ct < (counts ? table + 1 : c->n_ct); ++ct) {
        if (c->tab_rows_serial[ct] == c->count_serial) continue;
        if (int rc = run_export_rows(c, ct, c->tab_keys[ct], c->tab_refs[ct], c->tab_rows[ct])) return rc;
        c->tab_rows_serial[ct] = c->count_serial;
    }
    FmtArgs a{};
    a.kind = counts ? K_COUNTS : table == LSG_TABLE_MERGED ? K_MERGED : table == LSG_TABLE_STEP1 ? K_STEP1 : table == LSG_TABLE_STEP1_KEPT ? K_KEPT : K_STEP2;
    for (int q = 0; q < 3; ++q) { a.posset[q] = c->posset[q].keys.as<int64_t>(); a.n_posset[q] = c->posset[q].n; }
    a.ct = counts ? table : 0; a.n_ct = c->n_ct; a.n_contigs = c->n_contigs;
    a.n = counts ? c->n_rows[table] : c->n_sites;
    for (int ct = 0; ct < c->n_ct; ++ct) {
        a.keys[ct] = c->tab_keys[ct].as<int64_t>(); a.refs[ct] = c->tab_refs[ct].as<uint8_t>(); a.rows[ct] = c->tab_rows[ct].as<uint32_t>();
        a.n_rows[ct] = c->n_rows[ct];
    }
    a.sites = c->d_calls.as<SiteRec>(); a.cands = c->ws[WS_CALL_CANDS].as<CandCt>();
    a.ref_ptr = c->d_ref_ptrs.as<const uint8_t*>(); a.contig_len = c->d_contig_len.as<int64_t>();
    const char* names = c->tab_names.as<char>();
    a.contig_off = reinterpret_cast<const uint32_t*>(names); a.ct_off = reinterpret_cast<const uint32_t*>(names + c->tab_ct_off_at);
    a.contig_txt = names + c->tab_contig_txt_at; a.ct_txt = names + c->tab_ct_txt_at;
    const int32_t* order = reinterpret_cast<const int32_t*>(names + c->tab_order_at);
    c->tab_bytes[table] = -1;
    c->tab_scratch_table = -1;
    if (a.n == 0) { c->tab_bytes[table] = 0; return 0; }
    if (a.n >= (int64_t)1 << 31) { set_error("lsg_format_table: %lld rows", (long long)a.n); return -2; }
    // scratch: len[n + 1] | off[n + 1] | lo[n_contigs + 1] | shift[n_contigs]
    const size_t at_off = align_up((size_t)(a.n + 1) * 4, 256), at_lo = at_off + align_up((size_t)(a.n + 1) * 8, 256),
                 at_shift = at_lo + align_up((size_t)(c->n_contigs + 1) * 8, 256), total = at_shift + (size_t)(c->n_contigs + 1) * 8;
    if (c->tab_scratch.reserve(total)) return -1;
    char* sc = c->tab_scratch.as<char>();
    a.len = reinterpret_cast<uint32_t*>(sc);
    uint64_t* off = reinterpret_cast<uint64_t*>(sc + at_off);
    int64_t* lo = reinterpret_cast<int64_t*>(sc + at_lo);
    int64_t* shift = reinterpret_cast<int64_t*>(sc + at_shift);
    a.off = off; a.shift = shift;
    const unsigned blocks = (unsigned)((a.n + 1 + 255) / 256);
    hipLaunchKernelGGL(k_row_len, dim3(blocks), dim3(256), 0, st, a);
    {
        hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t*> in(a.len, Widen());
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, off, (int)(a.n + 1), st));
        if (c->d_cub_tmp.reserve(tb + 256)) return -1;
        tb = c->d_cub_tmp.cap;
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb, in, off, (int)(a.n + 1), st));
    }
    hipLaunchKernelGGL(k_contig_lo, dim3((unsigned)(c->n_contigs / 64 + 1)), dim3(64), 0, st, a, lo);
    hipLaunchKernelGGL(k_contig_shift, dim3(1), dim3(1), 0, st, lo, off, order, c->n_contigs, shift);
    LSG_HIP(hipMemcpyAsync(c->h_pin, off + a.n, 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    const int64_t bytes = (int64_t)c->h_pin[0];
    if (bytes > 0) {
        if (c->tab_text[table].reserve((size_t)bytes)) return -1;
        a.text = c->tab_text[table].as<char>();
        hipLaunchKernelGGL(k_row_put, dim3(blocks), dim3(256), 0, st, a);
        LSG_HIP(hipGetLastError());
        LSG_HIP(hipStreamSynchronize(st));
    }
    c->tab_bytes[table] = bytes;
    c->tab_scratch_table = table; c->tab_scratch_rows = a.n;      // (the rows' places stay in the scratch: lsg_step2_summary reads the text by them)
    if (n_bytes) *n_bytes = bytes;
    return 0;
}

// Column kinds of the step-2 text + its surviving rows as table LSG_TABLE_STEP3_ROWS
int run_step2_summary(lsg_ctx* c, int32_t n_cols, uint8_t* kinds, int64_t* n_survivor_bytes) {
    if (n_survivor_bytes) *n_survivor_bytes = 0;
    if (n_cols < 7 || n_cols > 64) { set_error("lsg_step2_summary: %d columns (7 to 64)", n_cols); return -2; }
    if (c->tab_bytes[LSG_TABLE_STEP2] < 0) { set_error("lsg_step2_summary: format table LSG_TABLE_STEP2 first"); return -2; }
    memset(kinds, 0, (size_t)n_cols);
    c->tab_bytes[LSG_TABLE_STEP3_ROWS] = -1;
    if (c->tab_bytes[LSG_TABLE_STEP2] == 0) { c->tab_bytes[LSG_TABLE_STEP3_ROWS] = 0; return 0; }
    if (c->tab_scratch_table != LSG_TABLE_STEP2 || c->tab_scratch_rows != c->n_sites || !c->called) { set_error("lsg_step2_summary: must follow the format of LSG_TABLE_STEP2 directly"); return -2; }
    hipStream_t st = c->stream;
    const int64_t n = c->tab_scratch_rows;
    const size_t at_off = align_up((size_t)(n + 1) * 4, 256), at_lo = at_off + align_up((size_t)(n + 1) * 8, 256),
                 at_shift = at_lo + align_up((size_t)(c->n_contigs + 1) * 8, 256);
    char* sc = c->tab_scratch.as<char>();
    TextRows t{c->tab_text[LSG_TABLE_STEP2].as<char>(), reinterpret_cast<const uint32_t*>(sc), reinterpret_cast<const uint64_t*>(sc + at_off),
               reinterpret_cast<const int64_t*>(sc + at_shift), c->d_calls.as<SiteRec>(), n};
    const int64_t* lo = reinterpret_cast<const int64_t*>(sc + at_lo);
    // second scratch: kinds[64] | len2[n + 1] | off2[n + 1] | shift2[n_contigs + 1]
    const size_t b_len = 256, b_off = b_len + align_up((size_t)(n + 1) * 4, 256), b_shift = b_off + align_up((size_t)(n + 1) * 8, 256), b_total = b_shift + (size_t)(c->n_contigs + 1) * 8;
    if (c->tab_scratch2.reserve(b_total)) return -1;
    char* s2 = c->tab_scratch2.as<char>();
    uint32_t* d_kinds = reinterpret_cast<uint32_t*>(s2);
    uint32_t* len2 = reinterpret_cast<uint32_t*>(s2 + b_len);
    uint64_t* off2 = reinterpret_cast<uint64_t*>(s2 + b_off);
    int64_t* shift2 = reinterpret_cast<int64_t*>(s2 + b_shift);
    LSG_HIP(hipMemsetAsync(d_kinds, 0, 256, st));
    const unsigned blocks = (unsigned)((n + 1 + 255) / 256);
    hipLaunchKernelGGL(k_text_kinds, dim3(blocks), dim3(256), 0, st, t, (int)n_cols, d_kinds);
    hipLaunchKernelGGL(k_text_survivors, dim3(blocks), dim3(256), 0, st, t, len2);
    {
        hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t*> in(len2, Widen());
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, off2, (int)(n + 1), st));
        if (c->d_cub_tmp.reserve(tb + 256)) return -1;
        tb = c->d_cub_tmp.cap;
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb, in, off2, (int)(n + 1), st));
    }
    const int32_t* order = reinterpret_cast<const int32_t*>(c->tab_names.as<char>() + c->tab_order_at);
    hipLaunchKernelGGL(k_contig_shift, dim3(1), dim3(1), 0, st, lo, off2, order, c->n_contigs, shift2);
    uint32_t h_kinds[64];
    LSG_HIP(hipMemcpyAsync(h_kinds, d_kinds, 256, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(c->h_pin, off2 + n, 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < n_cols; ++q) kinds[q] = (uint8_t)h_kinds[q];
    const int64_t bytes = (int64_t)c->h_pin[0];
    if (bytes > 0) {
        if (c->tab_text[LSG_TABLE_STEP3_ROWS].reserve((size_t)bytes)) return -1;
        hipLaunchKernelGGL(k_text_copy_rows, dim3(blocks), dim3(256), 0, st, t, len2, off2, shift2, c->tab_text[LSG_TABLE_STEP3_ROWS].as<char>());
        LSG_HIP(hipGetLastError());
        LSG_HIP(hipStreamSynchronize(st));
    }
    c->tab_bytes[LSG_TABLE_STEP3_ROWS] = bytes;
    if (n_survivor_bytes) *n_survivor_bytes = bytes;
    return 0;
}

static int table_ready(lsg_ctx* c, int32_t table, const char* who) {
    if (table < 0 || table >= LSG_TABLE_SLOTS || c->tab_bytes[table] < 0) { set_error("%s: table %d is not formatted (lsg_format_table)", who, table); return -2; }
    return 0;
}

int run_copy_table(lsg_ctx* c, int32_t table, char* dst, int64_t capacity) {
    if (int rc = table_ready(c, table, "lsg_copy_table")) return rc;
    const int64_t n = c->tab_bytes[table];
    if (capacity < n) { set_error("lsg_copy_table: capacity %lld < %lld bytes", (long long)capacity, (long long)n); return -2; }
    if (n == 0) return 0;
    LSG_HIP(hipMemcpyAsync(dst, c->tab_text[table].p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// device -> two pinned staging buffers -> pwrite at the file's end; the copy of piece i + 1 runs while piece i is written
int run_append_table(lsg_ctx* c, int32_t table, const char* path) {
    if (int rc = table_ready(c, table, "lsg_append_table")) return rc;
    const int64_t n = c->tab_bytes[table];
    const char* src = c->tab_text[table].as<char>();
    LSG_HIP(hipSetDevice(c->device));
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { set_error("lsg_append_table: cannot open %s", path); return -2; }
    off_t at = lseek(fd, 0, SEEK_END);
    if (n == 0) { close(fd); return 0; }
    const int64_t piece = (int64_t)32 << 20;
    hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; char* buf[2] = {nullptr, nullptr};
    int rc = 0;
    auto fail = [&](const char* what) { set_error("lsg_append_table: %s", what); rc = -1; };
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) fail("no stream");
    for (int b = 0; b < 2 && !rc; ++b)
        if (hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess || hipHostMalloc((void**)&buf[b], (size_t)std::min(piece, n), hipHostMallocDefault) != hipSuccess) fail("no pinned staging memory");
    const int64_t n_pieces = (n + piece - 1) / piece;
    auto issue = [&](int64_t i) {
        const int64_t b = i * piece, len = std::min(piece, n - b);
        if (hipMemcpyAsync(buf[i & 1], src + b, (size_t)len, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(ev[i & 1], st) != hipSuccess) fail("device to host copy failed");
    };
    if (!rc) issue(0);
    for (int64_t i = 0; i < n_pieces && !rc; ++i) {
        if (i + 1 < n_pieces) issue(i + 1);
        if (rc) break;
        if (hipEventSynchronize(ev[i & 1]) != hipSuccess) { fail("device to host copy failed"); break; }
        const int64_t b = i * piece, len = std::min(piece, n - b);
        int64_t done = 0;
        while (done < len) {
            const ssize_t w = pwrite(fd, buf[i & 1] + done, (size_t)(len - done), at + (off_t)(b + done));
            if (w <= 0) { fail("write failed"); break; }
            done += w;
        }
    }
    if (st) (void)hipStreamSynchronize(st);
    for (int b = 0; b < 2; ++b) { if (buf[b]) (void)hipHostFree(buf[b]); if (ev[b]) (void)hipEventDestroy(ev[b]); }
    if (st) (void)hipStreamDestroy(st);
    if (close(fd) != 0 && !rc) fail("write failed");
    return rc;
}

int run_free_table(lsg_ctx* c, int32_t table) {
    if (table >= LSG_TABLE_SLOTS) { set_error("lsg_free_table: no table %d", table); return -2; }
    for (int t = 0; t < LSG_TABLE_SLOTS; ++t)
        if (table < 0 || t == table) { c->tab_text[t].release(); c->tab_bytes[t] = -1; }
    if (table < 0) {
        for (int ct = 0; ct < LSG_MAX_CELLTYPES; ++ct) { c->tab_keys[ct].release(); c->tab_refs[ct].release(); c->tab_rows[ct].release(); c->tab_rows_serial[ct] = 0; }
        c->tab_scratch.release(); c->tab_scratch2.release(); c->tab_scratch_table = -1;
    }
    return 0;
}

} // namespace lsg
