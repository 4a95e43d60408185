// Host build of longsom_amd/csrc/bamrec_core.h + inflate_core.h (the GPU ingest's per-record and per-block code) against the host
// decoder: the BAM is inflated block by block with inflate_raw, its records are validated, looked up and walked with the device code
// (one "lane" and 64 emulated lanes), and every array and counter must equal what lsio_decode_bam returns for the same file.
//   usage: test_bamrec <bam> <barcodes file: one cleaned barcode per line> <min_mapq> <legacy 0|1>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../longsom_amd/csrc/inflate_core.h"
#include "../../longsom_amd/csrc/cbtable_host.h"

extern "C" {
typedef struct {
    int64_t n_reads, n_segs, n_events;
    int32_t* read_tid; int32_t* read_pos; uint16_t* read_flag; uint8_t* read_mapq; int32_t* read_cb;
    uint32_t* seg_read; int32_t* seg_start; int32_t* seg_len; int64_t* seg_ev_off; uint16_t* events;
    int32_t n_contigs; char* contig_names; int64_t* contig_len;
    int64_t total_reads, pass_reads, cb_not_found, cb_not_matched, mapq_filtered;
    int32_t n_barcodes; char* barcodes;
    int64_t n_tally; int64_t* cb_pass; int64_t* cb_low;
} lsio_decoded;
int lsio_decode_bam(const char* path, const char* barcodes, int32_t n_barcodes, const int32_t* ids, int32_t min_mapq, int32_t n_threads, lsio_decoded** out);
void lsio_set_legacy_del_merge(int on);
void lsio_free_decoded(lsio_decoded* d);
const char* lsio_last_error(void);
}

#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

int main(int argc, char** argv) {
    if (argc < 5) FAIL("usage");
    const int min_mapq = atoi(argv[3]), legacy = atoi(argv[4]);
    std::ifstream bf(argv[2]); std::string joined, line; int32_t nb = 0;
    while (std::getline(bf, line)) { if (line.empty()) continue; joined += line; joined += '\n'; ++nb; }
    lsio_set_legacy_del_merge(legacy);
    lsio_decoded* want = nullptr;
    if (lsio_decode_bam(argv[1], joined.c_str(), nb, nullptr, min_mapq, 1, &want) != 0) FAIL("host decode failed: %s", lsio_last_error());
    std::ifstream f(argv[1], std::ios::binary); std::stringstream ss; ss << f.rdbuf(); const std::string file = ss.str();
    const uint8_t* d = (const uint8_t*)file.data();
    // BGZF blocks -> one uncompressed stream through the device's inflate
    std::vector<uint8_t> u; std::vector<uint8_t> tab(lsi::T_SYM); std::vector<uint8_t> tlens(lsi::T_LENS); lsi::Tab t{tab.data(), tlens.data(), 1};
    for (size_t off = 0; off + 18 <= file.size();) {
        const uint32_t xlen = lsr::rd16(d + off + 10), bsize = lsr::rd16(d + off + 16) + 1u, usize = lsr::rd32(d + off + bsize - 4);
        const size_t at = u.size(); u.resize(at + usize);
        if (usize && lsi::inflate_raw(d + off + 12 + xlen, bsize - xlen - 20, u.data() + at, usize, t) != 0) FAIL("inflate_raw failed at %zu", off);
        off += bsize;
    }
    if (memcmp(u.data(), "BAM\1", 4) != 0) FAIL("no magic");
    size_t p = 8 + lsr::rd32(u.data() + 4);
    const uint32_t n_ref = lsr::rd32(u.data() + p); p += 4;
    std::vector<int64_t> lens;
    for (uint32_t i = 0; i < n_ref; ++i) { const uint32_t ln = lsr::rd32(u.data() + p); p += 4 + ln; lens.push_back(lsr::rd32(u.data() + p)); p += 4; }
    lsr::CbTableHost cbt; cbt.build(joined.c_str(), nb, nullptr);
    const lsr::CbTable tv = cbt.view();
    int64_t total = 0, pass = 0, nf = 0, nm = 0, lowq = 0;
    std::vector<int64_t> cb_pass((size_t)cbt.n_tally, 0), cb_low((size_t)cbt.n_tally, 0);
    std::vector<int32_t> r_tid, r_pos, r_cb; std::vector<uint16_t> r_flag; std::vector<uint8_t> r_mapq;
    std::vector<uint32_t> s_read; std::vector<int32_t> s_start, s_len; std::vector<int64_t> s_off; std::vector<uint16_t> ev, ev64;
    while (p + 4 <= u.size()) {
        const uint32_t bs = lsr::rd32(u.data() + p); const uint8_t* rec = u.data() + p + 4;
        if (p + 4 + bs > u.size()) FAIL("truncated record");
        if (lsr::validate(rec, bs, (int32_t)n_ref, lens.data()) != lsr::REC_OK) FAIL("record failed validation");
        p += 4 + bs;
        const int32_t tid = (int32_t)lsr::rd32(rec); const uint32_t mapq = rec[9], n_cigar = lsr::rd16(rec + 12), flag = lsr::rd16(rec + 14);
        if (tid < 0) continue;
        ++total;
        uint32_t cb = 0, raw = 0, clean = 0;
        if (!lsr::find_cb(rec, bs, &cb, &raw, &clean)) { ++nf; continue; }
        const int32_t id = lsr::cb_lookup(tv, rec + cb, clean);
        if (id < 0) { ++nm; continue; }
        if ((int)mapq < min_mapq) { ++lowq; ++cb_low[(size_t)id]; } else { ++pass; ++cb_pass[(size_t)id]; }
        if ((flag & 0x4) || n_cigar == 0) continue;
        const uint32_t r = (uint32_t)r_tid.size();
        r_tid.push_back(tid); r_pos.push_back((int32_t)lsr::rd32(rec + 4)); r_flag.push_back((uint16_t)((flag & 0x0fffu) | (clean < raw ? 0x8000u : 0u)));
        r_mapq.push_back((uint8_t)mapq); r_cb.push_back(id);
        const lsr::Shape sh = lsr::walk<false>(rec, legacy, 0, 1, r, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
        const size_t s0 = s_read.size(), e0 = ev.size();
        s_read.resize(s0 + sh.n_segs); s_start.resize(s0 + sh.n_segs); s_len.resize(s0 + sh.n_segs); s_off.resize(s0 + sh.n_segs);
        ev.resize(e0 + sh.n_events, 0xDEAD); ev64.resize(e0 + sh.n_events, 0xDEAD);
        const lsr::Shape sh2 = lsr::walk<true>(rec, legacy, 0, 1, r, s_read.data() + s0, s_start.data() + s0, s_len.data() + s0, s_off.data() + s0, (int64_t)e0, ev.data());
        if (sh2.n_segs != sh.n_segs || sh2.n_events != sh.n_events) FAIL("count and emit passes disagree");
        std::vector<uint32_t> a(sh.n_segs); std::vector<int32_t> b(sh.n_segs), c(sh.n_segs); std::vector<int64_t> o(sh.n_segs);
        for (uint32_t lane = 0; lane < 64; ++lane)              // a wave's 64 lanes, one after another
            lsr::walk<true>(rec, legacy, lane, 64, r, a.data(), b.data(), c.data(), o.data(), (int64_t)e0, ev64.data());
        for (uint32_t i = 0; i < sh.n_segs; ++i)
            if (a[i] != s_read[s0 + i] || b[i] != s_start[s0 + i] || c[i] != s_len[s0 + i] || o[i] != s_off[s0 + i]) FAIL("64-lane walk: segments differ");
    }
    if (ev != ev64) FAIL("64-lane walk: events differ");
    if (total != want->total_reads || pass != want->pass_reads || nf != want->cb_not_found || nm != want->cb_not_matched || lowq != want->mapq_filtered)
        FAIL("counters differ: %ld %ld %ld %ld %ld vs %ld %ld %ld %ld %ld", (long)total, (long)pass, (long)nf, (long)nm, (long)lowq,
             (long)want->total_reads, (long)want->pass_reads, (long)want->cb_not_found, (long)want->cb_not_matched, (long)want->mapq_filtered);
    if ((int64_t)r_tid.size() != want->n_reads || (int64_t)s_read.size() != want->n_segs || (int64_t)ev.size() != want->n_events) FAIL("shapes differ");
#define CMP(v, w, n) if ((n) && memcmp((v).data(), (w), (size_t)(n) * sizeof((v)[0])) != 0) FAIL("array " #v " differs")
    CMP(r_tid, want->read_tid, want->n_reads); CMP(r_pos, want->read_pos, want->n_reads); CMP(r_flag, want->read_flag, want->n_reads);
    CMP(r_mapq, want->read_mapq, want->n_reads); CMP(r_cb, want->read_cb, want->n_reads);
    CMP(s_read, want->seg_read, want->n_segs); CMP(s_start, want->seg_start, want->n_segs); CMP(s_len, want->seg_len, want->n_segs);
    CMP(s_off, want->seg_ev_off, want->n_segs); CMP(ev, want->events, want->n_events);
    for (int64_t i = 0; i < want->n_tally; ++i) if (cb_pass[(size_t)i] != want->cb_pass[i] || cb_low[(size_t)i] != want->cb_low[i]) FAIL("per-barcode tallies differ");
    printf("bamrec ok: %ld records -> %ld reads, %ld segments, %ld events equal the host decoder's\n", (long)total, (long)want->n_reads, (long)want->n_segs, (long)want->n_events);
    lsio_free_decoded(want);
    return 0;
}
