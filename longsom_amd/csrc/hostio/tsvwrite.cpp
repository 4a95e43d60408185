// Native writers of the three big tables of the SNV chain (liblongsom_io.so): the BaseCellCounter TSV
// (BaseCellCounter.py:300-308), the merged TSV (MergeBaseCellCounts.py:59-84,116-204) and the step-1 TSV
// (BaseCellCalling.step1.py:430-476).  At 24 M sites these are tens of GB of text; formatting them in Python took
// 26 s for a 300 k-read sample and dominated the end-to-end run by two orders of magnitude over the GPU work.
// Rows only: the caller (longsom_amd/tsvio.py) writes the header lines, these functions append.  Output order is the
// reference's: contigs in Python string order, positions ascending.  Sites are formatted in parallel chunks and
// written in order.  The text rules mirror tsvio.py's Python formatters (kept as the test reference).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>

#include "../../../include/longsom_hip.h"

namespace {

const char* const INFO_FIELD = "DP|NC|CC|BC|BQ|BCf|BCr";
const char* const CT_FILTER_NAMES[7] = {"", "Non-Significant", "Low-Significance", "Multi-allelic", "Low_cells", "Low_reads", "PASS"};
const struct { uint32_t bit; const char* name; } SITE_FILTER_NAMES[7] = {
    {1, "Multiple_cell_types"}, {2, "Multi-allelic"}, {4, "Min_cell_types"}, {8, "Cell_type_noise"},
    {16, "Noisy_site"}, {32, "LC_Upstream"}, {64, "LC_Downstream"}};
const uint32_t SF_CANDIDATE = 1u << 31;

thread_local char g_err[256] = "";
void set_err(const char* m) { snprintf(g_err, sizeof(g_err), "%s", m); }

// threads a writer starts by default: the cores this process may really use (its cgroup's CPU quota: the GPU box shows 256 cores and
// grants 16), at most 32
int default_threads() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32]; long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long long c = (atoll(q) + per - 1) / per; if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
        fclose(f);
    }
    return (int)std::min(32u, n);
}

// decimal digits two at a time, written backwards into a small buffer; ONE append per number (per row in put_row): a push_back per
// character (a capacity check each) was what the 16 GB of tables at C2 spent most of their time in
static const char DIG2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                           "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline char* fmt_u64(char* p, uint64_t v) {            // writes the digits of v at p, returns the end
    char b[24]; int n = 0;
    while (v >= 100) { const uint64_t q = v / 100; const unsigned r = (unsigned)(v - q * 100); b[n++] = DIG2[2 * r + 1]; b[n++] = DIG2[2 * r]; v = q; }
    if (v >= 10) { b[n++] = DIG2[2 * v + 1]; b[n++] = DIG2[2 * v]; } else b[n++] = (char)('0' + v);
    while (n) *p++ = b[--n];
    return p;
}
inline void put_u64(std::string& s, uint64_t v) { char b[24]; s.append(b, (size_t)(fmt_u64(b, v) - b)); }
inline void put_i64(std::string& s, int64_t v) { if (v < 0) { s.push_back('-'); put_u64(s, (uint64_t)(-v)); } else put_u64(s, (uint64_t)v); }

// 'DP|NC|CC|BC|BQ|BCf|BCr' values of one 42-word row: six printed classes per vector (32 numbers of at most 10 digits and their separators)
inline void put_row(std::string& s, const uint32_t* c) {
    char buf[32 * 11 + 8]; char* p = buf;
    p = fmt_u64(p, c[0]); *p++ = '|'; p = fmt_u64(p, c[1]);
    for (int o : {2, 10, 18, 26, 34}) {
        *p++ = '|';
        for (int k = 0; k < 6; ++k) { if (k) *p++ = ':'; p = fmt_u64(p, c[o + k]); }
    }
    s.append(buf, (size_t)(p - buf));
}

// repr(k / 10000.0) for the integer k = round(p, 4) * 1e4 (0 <= k): shortest text, at least one decimal
inline void put_p4(std::string& s, int64_t k) {
    if (k < 0) { s.push_back('-'); k = -k; }
    put_u64(s, (uint64_t)(k / 10000));
    s.push_back('.');
    int f = (int)(k % 10000);
    char d[4] = {(char)('0' + f / 1000), (char)('0' + f / 100 % 10), (char)('0' + f / 10 % 10), (char)('0' + f % 10)};
    int n = 4; while (n > 1 && d[n - 1] == '0') --n;
    s.append(d, (size_t)n);
}

// str(round(a / float(b), 4)): the exact binary quotient correctly rounded (half-even) to 4 decimals, trailing zeros cut
inline void put_ratio(std::string& s, int64_t a, int64_t b) {
    char buf[64];
    const double x = (double)a / (double)b;
    int n = snprintf(buf, sizeof(buf), "%.4f", x);              // glibc rounds the exact binary value correctly
    while (n > 0 && buf[n - 1] == '0' && buf[n - 2] != '.') --n;
    s.append(buf, (size_t)n);
}

std::vector<std::string> split_lines(const char* joined, int n) {
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

// contigs in Python string order
std::vector<int> contig_order(const std::vector<std::string>& names) {
    std::vector<int> o(names.size());
    for (size_t i = 0; i < o.size(); ++i) o[i] = (int)i;
    std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return names[(size_t)a] < names[(size_t)b]; });
    return o;
}

// [lo, hi) of the keys (sorted by (tid, pos)) that belong to contig tid
inline void contig_range(const int64_t* keys, int64_t n, int tid, int64_t& lo, int64_t& hi) {
    lo = std::lower_bound(keys, keys + n, (int64_t)tid << 32) - keys;
    hi = std::lower_bound(keys, keys + n, ((int64_t)tid + 1) << 32) - keys;
}

// Runs fmt(i, text) for i in [0, n) over `order` (a permutation of site indices in output order) in parallel chunks and
// appends the texts to `path` in order.
template <class F>
int write_chunks(const char* path, const std::vector<int64_t>& order, int n_threads, F fmt) {
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { set_err("cannot open the output file"); return -1; }
    off_t at = lseek(fd, 0, SEEK_END);                  // (the caller has written the header lines: rows are appended)
    const int64_t n = (int64_t)order.size();
    const int T = n_threads > 0 ? n_threads : default_threads();
    const int64_t CH = 16384;
    const int64_t n_chunks = (n + CH - 1) / CH;
    // A batch of chunks is formatted by the threads; their sizes give every chunk its place in the file, and the same threads write their
    // chunks there at once (pwrite): the copy into the page cache runs on every thread, not on one writer (2.6 GB/s for the 16 GB of C2's tables).
    std::atomic<bool> ok{true};
    bool use_mmap = getenv("LONGSOM_TABLES_MMAP") != nullptr;      // (measured on the GPU box: the page faults of a shared mapping cost more than the inode lock of pwrite; kept for file systems where they do not)
    for (int64_t c0 = 0; c0 < n_chunks && ok; c0 += (int64_t)T * 4) {
        const int64_t c1 = std::min(n_chunks, c0 + (int64_t)T * 4);
        std::vector<std::string> text((size_t)(c1 - c0));
        std::vector<off_t> where((size_t)(c1 - c0));
        auto run = [&](auto&& body) {
            std::atomic<int64_t> next(c0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&]() { for (int64_t c = next.fetch_add(1); c < c1; c = next.fetch_add(1)) body(c); });
            for (auto& t : th) t.join();
        };
        run([&](int64_t c) {
            std::string& s = text[(size_t)(c - c0)];
            s.reserve(1u << 22);
            const int64_t e = std::min(n, (c + 1) * CH);
            for (int64_t i = c * CH; i < e; ++i) fmt(order[(size_t)i], s);
        });
        const off_t batch_at = at;
        for (size_t i = 0; i < text.size(); ++i) { where[i] = at; at += (off_t)text[i].size(); }
        // Buffered writes to ONE file serialise on its inode (one thread's copy into the page cache: ~1 GB/s on the GPU box); with
        // LONGSOM_TABLES_MMAP=1 the pages of a shared mapping are faulted in and filled by every thread at once instead (slower there).
        char* map = nullptr; size_t map_len = 0; off_t map_off = 0;
        if (use_mmap && at > batch_at && ftruncate(fd, at) == 0) {
            const long pg = sysconf(_SC_PAGESIZE);
            map_off = batch_at / pg * pg; map_len = (size_t)(at - map_off);
            void* m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, map_off);
            if (m != MAP_FAILED) map = (char*)m; else use_mmap = false;
        } else if (at > batch_at) use_mmap = false;
        run([&](int64_t c) {
            const std::string& s = text[(size_t)(c - c0)];
            if (map) { memcpy(map + (where[(size_t)(c - c0)] - map_off), s.data(), s.size()); return; }
            size_t done = 0;
            while (done < s.size()) {
                const ssize_t w = pwrite(fd, s.data() + done, s.size() - done, where[(size_t)(c - c0)] + (off_t)done);
                if (w <= 0) { ok = false; return; }
                done += (size_t)w;
            }
        });
        if (map && munmap(map, map_len) != 0) ok = false;
    }
    if (close(fd) != 0) ok = false;
    if (!ok) { set_err("write failed"); return -1; }
    return 0;
}

// the merged site list of n_ct sorted key arrays: site keys and, per cell type, the row index or -1
struct Joined { std::vector<int64_t> key; std::vector<int64_t> row[LSG_MAX_CELLTYPES]; };
Joined outer_join(int n_ct, const int64_t* const* keys, const int64_t* n) {
    Joined j;
    int64_t p[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};
    while (true) {
        int64_t k = INT64_MAX;
        for (int c = 0; c < n_ct; ++c) if (p[c] < n[c] && keys[c][p[c]] < k) k = keys[c][p[c]];
        if (k == INT64_MAX) break;
        j.key.push_back(k);
        for (int c = 0; c < n_ct; ++c) {
            if (p[c] < n[c] && keys[c][p[c]] == k) j.row[c].push_back(p[c]++); else j.row[c].push_back(-1);
        }
    }
    return j;
}

std::vector<int64_t> output_order(const int64_t* keys, int64_t n, const std::vector<std::string>& names) {
    std::vector<int64_t> order; order.reserve((size_t)n);
    for (int tid : contig_order(names)) {
        int64_t lo, hi; contig_range(keys, n, tid, lo, hi);
        for (int64_t i = lo; i < hi; ++i) order.push_back(i);
    }
    return order;
}

bool keys_ok(const int64_t* keys, int64_t n, int n_contigs) {
    for (int64_t i = 0; i < n; ++i) {
        if (keys[i] < 0 || (keys[i] >> 32) >= n_contigs) return false;
        if (i && keys[i] <= keys[i - 1]) return false;
    }
    return true;
}

} // namespace

extern "C" {

const char* lsio_tsv_last_error(void) { return g_err; }

// n bytes from src to dst on the host's threads: the native tables come back to Python as bytes objects of gigabytes (the kept rows of a
// 10 M-read step-1 table: 2.7 GB), and one thread's memcpy into fresh pages moves ~2-3 GB/s
int lsio_copy_bytes(char* dst, const char* src, int64_t n, int32_t n_threads) {
    if (n <= 0) return 0;
    int T = n_threads > 0 ? n_threads : default_threads();
    if (n < (int64_t)(8 << 20)) T = 1;
    if (T <= 1) { memcpy(dst, src, (size_t)n); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, (size_t)(hi - lo)); });
    }
    for (auto& x : th) x.join();
    return 0;
}

// A whole file from a buffer, the copy into the page cache spread over the threads (a shared mapping of the file: buffered writes to one
// file serialise on its inode); the step-2 table of C2 is 2.7 GB.  Replaces the file.
int lsio_write_bytes(const char* path, const char* data, int64_t n, int32_t n_threads) {
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_err("cannot open the output file"); return -1; }
    bool ok = true;
    if (n > 0) {
        void* m = (getenv("LONGSOM_TABLES_MMAP") != nullptr && ftruncate(fd, (off_t)n) == 0) ? mmap(nullptr, (size_t)n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
        if (m != MAP_FAILED) {
            const int T = n_threads > 0 ? n_threads : default_threads();
            const int64_t piece = std::max<int64_t>(1 << 22, (n + T - 1) / T);
            std::vector<std::thread> th;
            for (int64_t a = 0; a < n; a += piece) th.emplace_back([=]() { memcpy((char*)m + a, data + a, (size_t)std::min(piece, n - a)); });
            for (auto& t : th) t.join();
            if (munmap(m, (size_t)n) != 0) ok = false;
        } else {
            int64_t done = 0;
            while (done < n) { const ssize_t w = pwrite(fd, data + done, (size_t)std::min<int64_t>(n - done, 1 << 30), (off_t)done); if (w <= 0) { ok = false; break; } done += w; }
        }
    }
    if (close(fd) != 0) ok = false;
    if (!ok) { set_err("write failed"); return -1; }
    return 0;
}
void lsio_free_text(char* p) { free(p); }

int lsio_write_count_rows(const char* path, const char* contig_names, int32_t n_contigs, const int64_t* keys, const uint8_t* refs,
                          const uint32_t* counts, int64_t n, int32_t n_threads) {
    const auto names = split_lines(contig_names, n_contigs);
    if (!keys_ok(keys, n, n_contigs)) { set_err("keys must be strictly ascending (tid << 32 | pos0) of known contigs"); return -2; }
    const auto order = output_order(keys, n, names);
    return write_chunks(path, order, n_threads, [&](int64_t i, std::string& s) {
        const int64_t k = keys[i];
        s += names[(size_t)(k >> 32)]; s.push_back('\t'); put_u64(s, (uint64_t)(k & 0xFFFFFFFF) + 1); s.push_back('\t');
        s.push_back((char)refs[i]); s.push_back('\t'); s += INFO_FIELD; s.push_back('\t');
        put_row(s, counts + i * LSG_ROW_WORDS); s.push_back('\n');
    });
}

int lsio_write_merged_rows(const char* path, const char* contig_names, int32_t n_contigs, int32_t n_ct, const int64_t* const* keys,
                           const uint8_t* const* refs, const uint32_t* const* counts, const int64_t* n, int32_t n_threads) {
    if (n_ct < 1 || n_ct > LSG_MAX_CELLTYPES) { set_err("bad number of cell types"); return -2; }
    const auto names = split_lines(contig_names, n_contigs);
    for (int c = 0; c < n_ct; ++c) if (!keys_ok(keys[c], n[c], n_contigs)) { set_err("keys must be strictly ascending (tid << 32 | pos0) of known contigs"); return -2; }
    const Joined j = outer_join(n_ct, keys, n);
    const auto order = output_order(j.key.data(), (int64_t)j.key.size(), names);
    return write_chunks(path, order, n_threads, [&](int64_t i, std::string& s) {
        const int64_t k = j.key[(size_t)i];
        s += names[(size_t)(k >> 32)]; s.push_back('\t');
        put_u64(s, (uint64_t)(k & 0xFFFFFFFF) + 1); s.push_back('\t'); put_u64(s, (uint64_t)(k & 0xFFFFFFFF) + 1); s.push_back('\t');
        // sort_set (MergeBaseCellCounts.py:48-57): distinct REFs by decreasing count, first-seen order on ties
        char seen[LSG_MAX_CELLTYPES]; int cnt[LSG_MAX_CELLTYPES]; int ns = 0;
        for (int c = 0; c < n_ct; ++c) {
            const int64_t r = j.row[c][(size_t)i];
            if (r < 0) continue;
            const char b = (char)refs[c][r];
            int q = 0; while (q < ns && seen[q] != b) ++q;
            if (q == ns) { seen[ns] = b; cnt[ns++] = 1; } else ++cnt[q];
        }
        int idx[LSG_MAX_CELLTYPES] = {0, 1, 2, 3};
        std::stable_sort(idx, idx + ns, [&](int a, int b) { return cnt[a] > cnt[b]; });
        for (int q = 0; q < ns; ++q) { if (q) s.push_back('|'); s.push_back(seen[idx[q]]); }
        s.push_back('\t'); s += INFO_FIELD;
        for (int c = 0; c < n_ct; ++c) {
            s.push_back('\t');
            const int64_t r = j.row[c][(size_t)i];
            if (r < 0) s += "NA"; else put_row(s, counts[c] + r * LSG_ROW_WORDS);
        }
        s.push_back('\n');
    });
}

// Step-1 rows for EVERY merged site (calls[i] <-> i-th site of the outer join).  *cand_text receives (malloc) the rows
// step 2 keeps (ALT != "." and FILTER != ".", the awk filter of BaseCellCalling.step2.py:23), in output order.
int lsio_write_step1_rows(const char* path, const char* contig_names, int32_t n_contigs, int32_t n_ct, const char* celltype_names,
                          const lsg_call* calls, int64_t n_calls, const int64_t* const* keys, const uint32_t* const* counts, const int64_t* n,
                          int32_t n_threads, char** cand_text, int64_t* cand_len) {
    if (n_ct < 1 || n_ct > LSG_MAX_CELLTYPES) { set_err("bad number of cell types"); return -2; }
    const auto names = split_lines(contig_names, n_contigs);
    const auto ctn = split_lines(celltype_names, n_ct);
    for (int c = 0; c < n_ct; ++c) if (!keys_ok(keys[c], n[c], n_contigs)) { set_err("keys must be strictly ascending (tid << 32 | pos0) of known contigs"); return -2; }
    const Joined j = outer_join(n_ct, keys, n);
    if ((int64_t)j.key.size() != n_calls) { set_err("the call records do not cover the merged sites"); return -2; }
    for (int64_t i = 0; i < n_calls; ++i) if (calls[i].key != j.key[(size_t)i]) { set_err("call records and count rows disagree on the sites"); return -2; }
    const auto order = output_order(j.key.data(), n_calls, names);
    // candidate rows are collected per chunk through a side buffer keyed by the site's output rank
    std::vector<int64_t> rank((size_t)n_calls);
    for (size_t r = 0; r < order.size(); ++r) rank[(size_t)order[r]] = (int64_t)r;
    std::vector<std::string> cand_rows((size_t)((n_calls + 16383) / 16384));
    // (a chunk is formatted by exactly one thread: its candidate buffer needs no lock)
    const bool kept_only = path == nullptr || !*path;      // only the rows step 2 keeps are wanted: the others are not even formatted
    const bool collect = cand_text != nullptr;
    auto fmt_row = [&](int64_t i, std::string& s) {
        const lsg_call& c = calls[i];
        if (kept_only && !(c.site_filter & SF_CANDIDATE)) return;      // (a row without an ALT is dropped by step 2's awk filter, step2.py:23)
        const int64_t k = c.key;
        const size_t row_start = s.size();
        s += names[(size_t)(k >> 32)]; s.push_back('\t');
        put_u64(s, (uint64_t)(k & 0xFFFFFFFF) + 1); s.push_back('\t'); put_u64(s, (uint64_t)(k & 0xFFFFFFFF) + 1); s.push_back('\t');
        s.push_back((char)c.ref); s.push_back('\t');
        std::string up, down;
        for (int q = 0; q < 5 && c.up_ctx[q]; ++q) up.push_back((char)c.up_ctx[q]);
        if (up.empty()) { up = "."; down = "."; } else for (int q = 0; q < 5 && c.down_ctx[q]; ++q) down.push_back((char)c.down_ctx[q]);
        const uint32_t sf = c.site_filter;
        std::string rest[2];
        for (int q = 0; q < 2; ++q) {
            const int64_t s_alt = q ? c.sum_alts_cc : c.sum_alts_bc, s_tot = q ? c.sum_nc : c.sum_dp, pk = q ? c.noise_p_cc : c.noise_p_bc;
            put_i64(rest[q], s_alt); rest[q].push_back(';'); put_i64(rest[q], s_tot); rest[q].push_back(';');
            if (c.sum_alts_bc == 0) rest[q] += "1"; else if (pk == -2) rest[q] += "nan"; else put_p4(rest[q], pk);
        }
        bool is_cand_row;
        if (sf & SF_CANDIDATE) {
            std::string alts, cts, dps, ncs, bcs, ccs, vafs, mcfs, bcps, ccps, flt;
            std::string alt_of[LSG_MAX_CELLTYPES]; int n_alt_str = 0; bool any_pass = false; bool first = true;
            for (int ct = 0; ct < n_ct; ++ct) {
                if (!((c.has_cand >> ct) & 1)) continue;
                const int na = c.n_alt[ct];
                const uint32_t* row = counts[ct] + j.row[ct][(size_t)i] * LSG_ROW_WORDS;
                const int64_t dp = row[0], nc = row[1];
                if (!first) { for (std::string* x : {&alts, &cts, &dps, &ncs, &bcs, &ccs, &vafs, &mcfs, &bcps, &ccps, &flt}) x->push_back(','); }
                first = false;
                std::string a;
                for (int q = 0; q < na; ++q) {
                    if (q) { for (std::string* x : {&a, &bcs, &ccs, &vafs, &mcfs, &bcps, &ccps}) x->push_back('|'); }
                    a.push_back("ACTG"[c.alt[ct][q] & 3]);
                    put_u64(bcs, c.alt_bc[ct][q]); put_u64(ccs, c.alt_cc[ct][q]);
                    put_ratio(vafs, c.alt_bc[ct][q], dp); put_ratio(mcfs, c.alt_cc[ct][q], nc);
                    put_p4(bcps, c.p_bc[ct][q]); put_p4(ccps, c.p_cc[ct][q]);
                }
                alts += a; cts += ctn[(size_t)ct]; put_i64(dps, dp); put_i64(ncs, nc);
                const char* fn = CT_FILTER_NAMES[c.ct_filter[ct] < 7 ? c.ct_filter[ct] : 0];
                flt += fn; if (!strcmp(fn, "PASS")) any_pass = true;
                bool dup = false; for (int q = 0; q < n_alt_str; ++q) dup |= alt_of[q] == a;
                if (!dup) alt_of[n_alt_str++] = a;
            }
            std::string FILTER;
            for (const auto& e : SITE_FILTER_NAMES) if (sf & e.bit) { if (!FILTER.empty()) FILTER.push_back(','); FILTER += e.name; }
            if (FILTER.empty()) FILTER = any_pass ? std::string("PASS") : flt;
            s += alts; s.push_back('\t'); s += FILTER; s.push_back('\t'); s += cts; s.push_back('\t'); s += up; s.push_back('\t'); s += down; s.push_back('\t');
            put_u64(s, (uint64_t)n_alt_str);
            for (const std::string* x : {&dps, &ncs, &bcs, &ccs, &vafs, &mcfs, &bcps, &ccps}) { s.push_back('\t'); s += *x; }
            s.push_back('\t'); put_i64(s, c.cell_types_min); s.push_back('\t'); put_i64(s, c.cell_types_min);
            s.push_back('\t'); s += rest[0]; s.push_back('\t'); s += rest[1]; s += "\t.\t"; s += flt;
            is_cand_row = alts != "." && FILTER != ".";
        } else {
            const bool noisy = (sf & 16) != 0;
            s += ".\t"; s += noisy ? "Noisy_site" : "."; s += "\t.\t"; s += up; s.push_back('\t'); s += down;
            s += "\t.\t.\t.\t.\t.\t.\t.\t.\t.\t"; put_i64(s, c.cell_types_min); s.push_back('\t'); put_i64(s, c.cell_types_min);
            s.push_back('\t'); s += rest[0]; s.push_back('\t'); s += rest[1]; s += "\t.\t.";
            is_cand_row = false;                                             // ALT is "."
        }
        s.push_back('\t'); s += INFO_FIELD;
        for (int ct = 0; ct < n_ct; ++ct) {
            s.push_back('\t');
            const int64_t r = j.row[ct][(size_t)i];
            if (r < 0) s += "NA"; else put_row(s, counts[ct] + r * LSG_ROW_WORDS);
        }
        s.push_back('\n');
        if (kept_only) { if (!is_cand_row) s.resize(row_start); return; }
        if (is_cand_row && collect) cand_rows[(size_t)(rank[(size_t)i] / 16384)].append(s, row_start, std::string::npos);
    };
    if (kept_only) {
        // the kept rows alone, chunk by chunk on the threads, straight into the chunks' candidate buffers
        const int T = n_threads > 0 ? n_threads : default_threads();
        std::atomic<int64_t> next(0);
        const int64_t n_chunks = (n_calls + 16383) / 16384;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                for (int64_t ch = next.fetch_add(1); ch < n_chunks; ch = next.fetch_add(1)) {
                    std::string& s = cand_rows[(size_t)ch];
                    const int64_t e = std::min<int64_t>(n_calls, (ch + 1) * 16384);
                    for (int64_t r = ch * 16384; r < e; ++r) fmt_row(order[(size_t)r], s);
                }
            });
        for (auto& t : th) t.join();
    } else {
        const int rc = write_chunks(path, order, n_threads, fmt_row);
        if (rc) return rc;
    }
    if (cand_text) {
        size_t tot = 0; for (auto& r : cand_rows) tot += r.size();
        char* out = (char*)malloc(tot + 1);
        if (!out) { set_err("out of memory"); return -1; }
        size_t o = 0; for (auto& r : cand_rows) { memcpy(out + o, r.data(), r.size()); o += r.size(); }
        out[tot] = 0;
        *cand_text = out; if (cand_len) *cand_len = (int64_t)tot;
    }
    return 0;
}

} // extern "C"
