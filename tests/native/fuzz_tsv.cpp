// CPU fuzz driver of the native text parsers of steps 2 and 3 (longsom_amd/csrc/hostio/tsvscan.cpp, tsvstep3.cpp), built with
// -fsanitize=address,undefined by tests/test_tsv_fuzz_cpu.py.  Takes a step-2 table, and for N seeded iterations damages its rows —
// bit flips, truncation, tabs / newlines / NULs written or removed, fields emptied or replaced by numbers and missing-value strings —
// then runs lsio_scan_rows, lsio_gather_lines (with and without NA blanking) and lsio_step3_rows on the result.  Every call must
// return (0, 1 = "not mine", or an error code); the sanitizers turn any read past a buffer into a failure of this program.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
struct lsio_row_scan { int64_t n_rows, n_comment_lines; int64_t* off; int64_t* key; int32_t* len; int32_t* filt_off; int32_t* filt_len; uint32_t* flags; };
int lsio_scan_rows(const char*, int64_t, const char*, int32_t, const char*, const char*, int32_t, const char*, int32_t, lsio_row_scan*);
void lsio_free_row_scan(lsio_row_scan*);
int lsio_gather_lines(const char*, const int64_t*, const int32_t*, int64_t, int32_t, int32_t, char**, int64_t*, int64_t*);
int lsio_step3_rows(const char*, int64_t, int32_t, const int32_t*, double, double, int64_t, int64_t, int64_t, const uint8_t*, char**, int64_t*, char**, int64_t*);
int lsio_step3_column_kinds(const char*, int64_t, int32_t, uint8_t*);
void lsio_free_text(char*);
}

static uint64_t rng_state = 1;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: fuzz_tsv step2.tsv iterations seed\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::string all; { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, n); } fclose(f);
    const int iters = atoi(argv[2]); rng_state = strtoull(argv[3], nullptr, 10) * 2654435761ull + 88172645463325252ull;
    // header and body
    std::vector<std::string> cols; std::string body;
    for (size_t a = 0; a < all.size();) {
        size_t e = all.find('\n', a); if (e == std::string::npos) e = all.size();
        const std::string line = all.substr(a, e - a);
        if (!line.empty() && line[0] == '#') { if (line.rfind("#CHROM", 0) == 0) { cols.clear(); for (size_t p = 0;;) { size_t t = line.find('\t', p); cols.push_back(line.substr(p, t == std::string::npos ? t : t - p)); if (t == std::string::npos) break; p = t + 1; } } }
        else if (!line.empty()) body += line + "\n";
        a = e + 1;
    }
    const char* want[15] = {"#CHROM", "Start", "REF", "ALT", "FILTER", "Cell_types", "Dp", "Nc", "Bc", "Cc", "VAF", "MCF", "Cell_type_Filter", "Cancer", "Non-Cancer"};
    int32_t col[15];
    for (int i = 0; i < 15; ++i) { col[i] = -1; for (size_t c = 0; c < cols.size(); ++c) if (cols[c] == want[i]) col[i] = (int32_t)c; }
    const char* subst[] = {"", "NA", "nan", "0", "-1", "007", "1e5", "0.5", "1.0", "A|C", "T", "N", "Cancer,Non-Cancer", "Non-Cancer", "PASS", "Multi-allelic", "99,100", "x|y|1:2|3:4", "\"q", "#"};
    int handled = 0, handed_back = 0, errors = 0;
    for (int it = 0; it < iters; ++it) {
        std::string t = body;
        const int n = 1 + (int)(rnd() % 12);
        for (int k = 0; k < n && !t.empty(); ++k) {
            const size_t at = rnd() % t.size();
            switch (rnd() % 7) {
                case 0: t[at] ^= (char)(1u << (rnd() % 8)); break;
                case 1: t[at] = '\t'; break;
                case 2: t[at] = '\n'; break;
                case 3: t[at] = 0; break;
                case 4: t.erase(at, 1 + rnd() % 40); break;
                case 5: { size_t b = t.rfind('\t', at); b = b == std::string::npos ? 0 : b + 1; size_t e = t.find_first_of("\t\n", at); if (e == std::string::npos) e = t.size(); if (e >= b) t.replace(b, e - b, subst[rnd() % (sizeof subst / sizeof *subst)]); break; }
                default: t.resize(at); break;
            }
        }
        lsio_row_scan sc;
        if (lsio_scan_rows(t.data(), (int64_t)t.size(), "chr1\nchr2\nchrM", 3, "Min|LR|gnomAD|LC|RNA", "Min_cell_types|Noisy_site|PoN", (int32_t)(rnd() % 9) - 1, "Non-Cancer", 1 + (int)(rnd() % 3), &sc) == 0) {
            for (int blank = 0; blank < 2; ++blank) {
                char* out = nullptr; int64_t len = 0; std::vector<int64_t> noff((size_t)sc.n_rows + 1);
                if (lsio_gather_lines(t.data(), sc.off, sc.len, sc.n_rows, blank, 1 + (int)(rnd() % 3), &out, &len, noff.data()) == 0) lsio_free_text(out);
            }
            lsio_free_row_scan(&sc);
        }
        char *a = nullptr, *b = nullptr; int64_t na = 0, nb = 0;
        // the kinds of cell of every column over the whole (damaged) table, handed on as the dtypes' evidence every other time
        std::vector<uint8_t> kinds(cols.size(), 0);
        const bool have_kinds = lsio_step3_column_kinds(t.data(), (int64_t)t.size(), (int32_t)cols.size(), kinds.data()) == 0;
        const int rc = lsio_step3_rows(t.data(), (int64_t)t.size(), (int32_t)cols.size(), col, 0.05, 0.3, 3, 2, 1 + (int64_t)(rnd() % 20000),
                                       have_kinds && (rnd() & 1) ? kinds.data() : nullptr, &a, &na, &b, &nb);
        if (rc == 0) { ++handled; lsio_free_text(a); lsio_free_text(b); } else if (rc == 1) ++handed_back; else ++errors;
    }
    printf("fuzz_tsv: %d tables handled, %d handed back, %d errors\n", handled, handed_back, errors);
    return 0;
}
