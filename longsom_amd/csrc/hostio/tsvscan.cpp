// Row scanner of the step-1 / step-2 tables (host, multi-threaded): what BaseCellCalling.step2.py:23 (the awk filter), :40-57 (the
// position of every row) and BaseCellCalling.step3.py:41-84 (Cell_types != Non-Cancer, the FILTER patterns that drop a row) read from a
// row, found for every row of the text in one pass, so that the Python side of steps 2 and 3 (longsom_amd/calling.py) only touches the
// rows it changes or keeps.  Nothing here decides anything the Python path does not: tests/test_calling_cpu.py runs both and compares.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Row { int64_t off; int64_t key; int32_t len, filt_off, filt_len; uint32_t flags; };

enum : uint32_t { F_ALT_DOT = 1, F_FILTER_DOT = 2, F_PAT_A = 4, F_PAT_B = 8, F_CT_MATCH = 16, F_SHORT = 32, F_BAD_POS = 64, F_UNKNOWN_CHROM = 128 };

std::vector<std::string> split_alt(const char* s) {
    std::vector<std::string> v;
    if (!s) return v;
    const char* p = s;
    while (*p) { const char* q = strchr(p, '|'); if (!q) q = p + strlen(p); if (q > p) v.emplace_back(p, q); p = *q ? q + 1 : q; }
    return v;
}

bool contains_any(const char* f, size_t n, const std::vector<std::string>& pats) {
    for (const auto& p : pats)
        if (p.size() <= n && memmem(f, n, p.data(), p.size())) return true;
    return false;
}

struct Scan {
    const char* text; const std::unordered_map<std::string, int32_t>* tid_of; const std::vector<std::string>* pa; const std::vector<std::string>* pb;
    const char* ct_value; size_t ct_len; int32_t ct_col;
    std::vector<Row> rows; int64_t comments = 0;
    void run(int64_t lo, int64_t hi) {                  // [lo, hi): whole lines
        std::string chrom; int32_t last_tid = -2; std::string last_chrom;
        int64_t a = lo;
        while (a < hi) {
            const char* nl = (const char*)memchr(text + a, '\n', (size_t)(hi - a));
            const int64_t e = nl ? (nl - text) : hi;
            if (e > a) {
                if (text[a] == '#') ++comments;
                else {
                    Row r{a, 0, (int32_t)(e - a), 0, 0, 0};
                    // fields 0..max(5, ct_col)
                    const int want = std::max(5, ct_col);
                    int64_t fs = a; int fi = 0;
                    int64_t f_off[8] = {0}; int32_t f_len[8] = {0};
                    int64_t ct_off = -1; int32_t ctl = 0;
                    while (fi <= want) {
                        const char* tb = (const char*)memchr(text + fs, '\t', (size_t)(e - fs));
                        const int64_t fe = tb ? (tb - text) : e;
                        if (fi < 8) { f_off[fi] = fs; f_len[fi] = (int32_t)(fe - fs); }
                        if (fi == ct_col) { ct_off = fs; ctl = (int32_t)(fe - fs); }
                        ++fi;
                        if (!tb) break;
                        fs = fe + 1;
                    }
                    if (fi <= want) r.flags |= F_SHORT;
                    if (fi > 5) {
                        if (f_len[4] == 1 && text[f_off[4]] == '.') r.flags |= F_ALT_DOT;
                        if (f_len[5] == 1 && text[f_off[5]] == '.') r.flags |= F_FILTER_DOT;
                        r.filt_off = (int32_t)(f_off[5] - a); r.filt_len = f_len[5];
                        if (contains_any(text + f_off[5], (size_t)f_len[5], *pa)) r.flags |= F_PAT_A;
                        if (contains_any(text + f_off[5], (size_t)f_len[5], *pb)) r.flags |= F_PAT_B;
                    }
                    if (ct_off >= 0 && (size_t)ctl == ct_len && memcmp(text + ct_off, ct_value, ct_len) == 0) r.flags |= F_CT_MATCH;
                    int32_t tid;
                    if (last_tid != -2 && (size_t)f_len[0] == last_chrom.size() && memcmp(text + f_off[0], last_chrom.data(), last_chrom.size()) == 0) tid = last_tid;
                    else {
                        last_chrom.assign(text + f_off[0], (size_t)f_len[0]);
                        auto it = tid_of->find(last_chrom);
                        tid = it == tid_of->end() ? -1 : it->second;
                        last_tid = tid;
                    }
                    if (tid < 0) { r.flags |= F_UNKNOWN_CHROM; tid = 0x7FFFFFFF; }
                    int64_t pos = 0; bool ok = fi > 1 && f_len[1] > 0 && f_len[1] <= 18;
                    for (int32_t i = 0; ok && i < f_len[1]; ++i) {
                        const char c = text[f_off[1] + i];
                        if (c < '0' || c > '9') ok = false; else pos = pos * 10 + (c - '0');
                    }
                    if (!ok) r.flags |= F_BAD_POS;
                    r.key = ((int64_t)tid << 32) | (pos & 0xFFFFFFFFll);
                    if (pos > 0xFFFFFFFFll) r.flags |= F_BAD_POS;
                    rows.push_back(r);
                }
            }
            a = e + 1;
        }
    }
};

thread_local char g_scan_err[256];

}  // namespace

extern "C" {

struct lsio_row_scan {
    int64_t n_rows, n_comment_lines;
    int64_t* off; int64_t* key; int32_t* len; int32_t* filt_off; int32_t* filt_len; uint32_t* flags;
};

const char* lsio_scan_last_error(void) { return g_scan_err; }

void lsio_free_row_scan(lsio_row_scan* s) {
    if (!s) return;
    free(s->off); free(s->key); free(s->len); free(s->filt_off); free(s->filt_len); free(s->flags);
    memset(s, 0, sizeof *s);
}

// text[0, n_bytes): a table whose rows are '\n'-terminated lines of tab-separated fields (CHROM Start End REF ALT FILTER ...); lines
// starting with '#' are counted, not returned.  contig_names: '\n'-joined.  patterns_a / patterns_b: '|'-separated literal strings
// searched in FILTER.  ct_col: the field compared with ct_value (-1: none).  flags per row: see F_* above.
int lsio_scan_rows(const char* text, int64_t n_bytes, const char* contig_names, int32_t n_contigs, const char* patterns_a, const char* patterns_b,
                   int32_t ct_col, const char* ct_value, int32_t threads, lsio_row_scan* out) {
    memset(out, 0, sizeof *out);
    if (n_bytes < 0 || ct_col > 7) { snprintf(g_scan_err, sizeof g_scan_err, "lsio_scan_rows: bad arguments"); return -1; }
    std::unordered_map<std::string, int32_t> tid_of;
    {
        const char* p = contig_names;
        for (int32_t i = 0; i < n_contigs && p; ++i) {
            const char* q = strchr(p, '\n');
            std::string name = q ? std::string(p, q) : std::string(p);
            tid_of[name] = i;                           // the last of two equal names wins: {n: i for i, n in enumerate(names)}
            p = q ? q + 1 : nullptr;
        }
    }
    const auto pa = split_alt(patterns_a), pb = split_alt(patterns_b);
    if (threads <= 0) threads = (int32_t)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n_bytes < (1 << 20)) threads = 1;
    std::vector<Scan> parts((size_t)threads);
    std::vector<int64_t> cut((size_t)threads + 1, n_bytes);
    cut[0] = 0;
    for (int32_t t = 1; t < threads; ++t) {
        int64_t c = n_bytes * t / threads;
        if (c < cut[(size_t)t - 1]) c = cut[(size_t)t - 1];
        const char* nl = c < n_bytes ? (const char*)memchr(text + c, '\n', (size_t)(n_bytes - c)) : nullptr;
        cut[(size_t)t] = nl ? (nl - text) + 1 : n_bytes;
    }
    const size_t ctl = ct_value ? strlen(ct_value) : 0;
    std::vector<std::thread> th;
    for (int32_t t = 0; t < threads; ++t) {
        parts[(size_t)t] = Scan{text, &tid_of, &pa, &pb, ct_value ? ct_value : "", ctl, ct_value ? ct_col : -1, {}, 0};
        if (threads == 1) parts[0].run(cut[0], cut[1]);
        else th.emplace_back([&parts, &cut, t] { parts[(size_t)t].run(cut[(size_t)t], cut[(size_t)t + 1]); });
    }
    for (auto& x : th) x.join();
    int64_t n = 0, nc = 0;
    for (auto& p : parts) { n += (int64_t)p.rows.size(); nc += p.comments; }
    out->n_rows = n; out->n_comment_lines = nc;
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    out->off = (int64_t*)malloc(m * 8); out->key = (int64_t*)malloc(m * 8); out->len = (int32_t*)malloc(m * 4);
    out->filt_off = (int32_t*)malloc(m * 4); out->filt_len = (int32_t*)malloc(m * 4); out->flags = (uint32_t*)malloc(m * 4);
    if (!out->off || !out->key || !out->len || !out->filt_off || !out->filt_len || !out->flags) {
        lsio_free_row_scan(out); snprintf(g_scan_err, sizeof g_scan_err, "lsio_scan_rows: out of memory"); return -2;
    }
    int64_t i = 0;
    for (auto& p : parts)
        for (const Row& r : p.rows) {
            out->off[i] = r.off; out->key[i] = r.key; out->len[i] = r.len; out->filt_off[i] = r.filt_off; out->filt_len[i] = r.filt_len; out->flags[i] = r.flags;
            ++i;
        }
    return 0;
}

// The lines (off[i], len[i]) of text, each followed by '\n', one after the other in a malloc'd buffer (*out, *out_len; lsio_free_text).
// blank_na: a field other than the first that is exactly "NA" becomes empty — what pandas' read_csv / to_csv round trip does to the
// table in the reference's step 2 (BaseCellCalling.step2.py:96,226).  new_off[i] (n + 1 entries, optional): where line i starts in *out.
int lsio_gather_lines(const char* text, const int64_t* off, const int32_t* len, int64_t n, int32_t blank_na, int32_t threads, char** out, int64_t* out_len,
                      int64_t* new_off) {
    *out = nullptr; *out_len = 0;
    if (n < 0) { snprintf(g_scan_err, sizeof g_scan_err, "lsio_gather_lines: bad arguments"); return -1; }
    if (threads <= 0) threads = (int32_t)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 4096) threads = 1;
    std::vector<int64_t> olen((size_t)n + 1, 0);
    auto na_at = [&](const char* l, int32_t L, int32_t i) {          // a field "NA" starts at l[i] (i > 0: behind a tab)
        return i + 2 <= L && l[i] == 'N' && l[i + 1] == 'A' && (i + 2 == L || l[i + 2] == '\t');
    };
    auto each = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int32_t t = 0; t < threads; ++t) {
            const int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
            if (threads == 1) fn(lo, hi); else th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
        }
        for (auto& x : th) x.join();
    };
    each([&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const char* l = text + off[i]; const int32_t L = len[i];
            int64_t o = (int64_t)L + 1;
            if (blank_na)
                for (const char* tb = (const char*)memchr(l, '\t', (size_t)L); tb; tb = (const char*)memchr(tb + 1, '\t', (size_t)(l + L - tb - 1)))
                    if (na_at(l, L, (int32_t)(tb - l) + 1)) o -= 2;
            olen[(size_t)i + 1] = o;
        }
    });
    for (int64_t i = 0; i < n; ++i) olen[(size_t)i + 1] += olen[(size_t)i];
    const int64_t total = olen[(size_t)n];
    char* buf = (char*)malloc((size_t)std::max<int64_t>(total, 1));
    if (!buf) { snprintf(g_scan_err, sizeof g_scan_err, "lsio_gather_lines: out of memory (%lld bytes)", (long long)total); return -2; }
    each([&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const char* l = text + off[i]; const int32_t L = len[i];
            char* w = buf + olen[(size_t)i];
            if (!blank_na || olen[(size_t)i + 1] - olen[(size_t)i] == (int64_t)L + 1) { memcpy(w, l, (size_t)L); w += L; }
            else {
                int32_t a = 0;
                while (a < L) {
                    const char* tb = (const char*)memchr(l + a, '\t', (size_t)(L - a));
                    const int32_t e = tb ? (int32_t)(tb - l) : L;
                    if (!(a > 0 && na_at(l, L, a) && e == a + 2)) { memcpy(w, l + a, (size_t)(e - a)); w += e - a; }
                    if (tb) *w++ = '\t';
                    a = e + 1;
                }
            }
            *w = '\n';
        }
    });
    if (new_off) memcpy(new_off, olen.data(), ((size_t)n + 1) * 8);
    *out = buf; *out_len = total;
    return 0;
}

}  // extern "C"
