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
};
enum { K_COUNTS = 0, K_MERGED, K_STEP1, K_KEPT };

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

// one row of the step-1 table (tsvwrite.cpp lsio_write_step1_rows, from the compact records instead of their lsg_call expansion)
template <class S> __device__ void step1_row(S& s, const FmtArgs& a, int64_t i) {
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
        if (!any_site) { if (any_pass) LIT(s, "PASS"); else filters(); }
        s.ch('\t');
        column([&](int ct) { put_name(s, a.ct_txt, a.ct_off, ct); });
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
        if (r[ct] < 0) LIT(s, "NA"); else put_row(s, a.rows[ct] + r[ct] * LSG_ROW_WORDS);
    }
    s.ch('\n');
}

template <class S> __device__ __forceinline__ void any_row(S& s, const FmtArgs& a, int64_t i) {
    switch (a.kind) {
        case K_COUNTS: count_row(s, a, i); break;
        case K_MERGED: merged_row(s, a, i); break;
        case K_STEP1: step1_row(s, a, i); break;
        default: if (a.sites[i].site_filter & (uint32_t)LSG_SF_CANDIDATE) step1_row(s, a, i);      // (a candidate's ALT and FILTER are never ".")
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
    const bool counts = table < LSG_TABLE_MERGED;
    if (counts && table >= c->n_ct) { set_error("lsg_format_table: no cell type %d", table); return -2; }
    if (!counts && !c->called) { set_error("lsg_format_table: the merged and step-1 tables need lsg_call_step1 (its merged site list)"); return -2; }
    hipStream_t st = c->stream;
    // flat copies of the rows the table prints (kept until the next count)
    for (int ct = counts ? table : 0; ct < (counts ? table + 1 : c->n_ct); ++ct) {
        if (c->tab_rows_serial[ct] == c->count_serial) continue;
        if (int rc = run_export_rows(c, ct, c->tab_keys[ct], c->tab_refs[ct], c->tab_rows[ct])) return rc;
        c->tab_rows_serial[ct] = c->count_serial;
    }
    FmtArgs a{};
    a.kind = counts ? K_COUNTS : table == LSG_TABLE_MERGED ? K_MERGED : table == LSG_TABLE_STEP1 ? K_STEP1 : K_KEPT;
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
    if (n_bytes) *n_bytes = bytes;
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
        c->tab_scratch.release();
    }
    return 0;
}

} // namespace lsg
