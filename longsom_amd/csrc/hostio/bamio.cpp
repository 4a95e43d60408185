// liblongsom_io.so — host side of the hot path: BGZF/BAM decode into the read-record arrays of
// include/longsom_hip.h, and a BAM writer for the synthetic workload.  Plain C ABI for ctypes.
//
// What the decode replaces in the reference (all pysam/htslib, not in the reference tree):
//   reading + routing records    workflow/scripts/PreProcessing/SplitBamCellTypes.py:51-124
//   CIGAR walk of the pileup     htslib bam_plp / resolve_cigar2 behind bam.pileup(...)
//                                (workflow/scripts/SNVCalling/BaseCellCounter.py:190-191)
//   per-entry symbol + quality   pysam PileupColumn.get_query_sequences(add_indels=True) +
//                                EasyReadPileup (BaseCellCounter.py:152-180,214-216)
// The per-read event classification is documented in SURVEY.md §8a ("Event classification
// restated from CIGAR") and cross-checked against oracle/plp_oracle.c, which iterates columns the
// way bam_plp does.
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../synth_model.h"

namespace {
thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t* p) { return (int32_t)rd32(p); }

struct Block { size_t off; uint32_t csize, usize; size_t uoff; };

// symbol class of a 4-bit BAM base code: A C G T N -> 0 1 3 2 6, everything else ('=', IUPAC) -> NA
const uint8_t NT16_SYM[16] = {15, 0, 1, 15, 3, 15, 15, 15, 2, 15, 15, 15, 15, 15, 15, 6};

struct Local {   // per-thread decode output
    std::vector<int32_t> read_tid, read_pos, read_cb; std::vector<uint16_t> read_flag; std::vector<uint8_t> read_mapq;
    std::vector<uint32_t> seg_read; std::vector<int32_t> seg_start, seg_len; std::vector<int64_t> seg_ev_off;
    std::vector<uint16_t> events;
    int64_t total = 0, pass = 0, cb_not_found = 0, cb_not_matched = 0, mapq = 0;
    std::vector<int64_t> cb_pass, cb_low;      // per dense barcode id: matched reads with MAPQ >= / < min_mapq (report of a later table)
};

// htslib <= 1.10: the last column of a D operation followed by another D is a deletion anchor too ("1D2D": 'D' where htslib >= 1.11
// gives 'O'); lsio_set_legacy_del_merge, default off = htslib >= 1.11
std::atomic<int> g_legacy_del_merge{0};
std::atomic<int> g_keep_unlisted{0};      // lsio_set_keep_unlisted: reads without a listed barcode stay in the arrays (cb = -1)

inline bool is_ref_op(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

// A record's own length fields must stay inside its block_size: name + CIGAR + packed sequence + qualities.  Everything
// that walks a record (decode_record, find_cb, the split writer) runs only on records that passed this test.
inline bool record_ok(const uint8_t* rec, uint32_t len) {
    if (len < 32) return false;
    const uint64_t l_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    return 32 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq <= (uint64_t)len;
}

// CB:Z value of a record, cleaned like barcode.split("-")[0]; false when the tag is missing.
bool find_cb(const uint8_t* rec, uint32_t len, const char** cb_out, size_t* len_out) {
    const uint32_t l_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    const uint8_t* aux = rec + 32 + l_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
    const uint8_t* end = rec + len;
    while (aux + 3 <= end) {
        const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
        aux += 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t* z = aux; while (z < end && *z) ++z; sz = (size_t)(z - aux) + 1;
                                  if (t0 == 'C' && t1 == 'B' && ty == 'Z' && z < end) {
                                      size_t clean = 0; while (clean < sz - 1 && aux[clean] != '-') ++clean;
                                      *cb_out = (const char*)aux; *len_out = clean; return true;
                                  } break; }
            case 'B': { if (aux + 5 > end) return false; const char st = (char)aux[0]; const uint32_t cnt = rd32(aux + 1);
                        sz = 5 + (size_t)cnt * ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4); break; }
            default: return false;
        }
        if (sz > (size_t)(end - aux)) return false;          // a field that claims more bytes than the record has
        aux += sz;
    }
    return false;
}

// Decode one BAM record (rec points at refID, i.e. after block_size) into L.
void decode_record(const uint8_t* rec, uint32_t len, const std::unordered_map<std::string, int32_t>& cbmap, int min_mapq, Local& L) {
    const int32_t tid = rdi32(rec), pos = rdi32(rec + 4);
    const uint32_t l_name = rec[8], mapq = rec[9], n_cigar = rd16(rec + 12), flag = rd16(rec + 14), l_seq = rd32(rec + 16);
    if (tid < 0) return;                              // infile.fetch() iterates reads placed on a reference
    ++L.total;
    const uint8_t* p = rec + 32 + l_name;
    const uint8_t* cigar = p; p += 4ull * n_cigar;
    const uint8_t* seq = p; p += (l_seq + 1) / 2;
    const uint8_t* qual = p; p += l_seq;
    const uint8_t* aux = p; const uint8_t* end = rec + len;
    // CB:Z tag (read.opt("CB"), SplitBamCellTypes.py:74-79)
    const char* cb = nullptr; size_t cb_len = 0;
    while (aux + 3 <= end) {
        const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
        aux += 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t* z = aux; while (z < end && *z) ++z; sz = (size_t)(z - aux) + 1;
                                  if (t0 == 'C' && t1 == 'B' && ty == 'Z' && z < end) { cb = (const char*)aux; cb_len = sz - 1; } break; }
            case 'B': { if (aux + 5 > end) { aux = end; continue; } const char st = (char)aux[0]; const uint32_t cnt = rd32(aux + 1);
                        const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; sz = 5 + es * cnt; break; }
            default: aux = end; continue;
        }
        if (sz > (size_t)(end - aux)) break;                  // a field that claims more bytes than the record has
        aux += sz;
    }
    // A read without a listed barcode is never counted, but it takes a place in the buffer of a pileup over the UNSPLIT BAM
    // (HCCVSingleCellGenotype.py:121-122: max_depth is about every read that passes the pileup's own filters): lsio_set_keep_unlisted
    // keeps it, with cb = -1, for the replay of that rule (genotype.hip)
    int32_t cb_id = -1; size_t clean = 0;
    if (!cb) ++L.cb_not_found;
    else {
        while (clean < cb_len && cb[clean] != '-') ++clean;                          // barcode.split("-")[0] (:83)
        auto it = cbmap.find(std::string(cb, clean));
        if (it == cbmap.end()) ++L.cb_not_matched;
        else {
            cb_id = it->second;
            if ((int)mapq < min_mapq) ++L.mapq; else ++L.pass;                       // report only; the device re-applies min_mq
            if ((size_t)cb_id < L.cb_pass.size()) { if ((int)mapq < min_mapq) ++L.cb_low[(size_t)cb_id]; else ++L.cb_pass[(size_t)cb_id]; }
        }
    }
    if (cb_id < 0 && !g_keep_unlisted.load(std::memory_order_relaxed)) return;
    if ((flag & 0x4) || n_cigar == 0) return;                                      // no alignment: nothing to pile up
    const uint32_t r = (uint32_t)L.read_tid.size();
    // SAM flags use 12 bits; bit 15 records that the raw CB carried a "-suffix" (the genotyping script looks the RAW tag up, lsg_genotype_cells)
    L.read_tid.push_back(tid); L.read_pos.push_back(pos); L.read_flag.push_back((uint16_t)((flag & 0x0fffu) | (cb_id >= 0 && clean < cb_len ? LSG_FLAG_CB_SUFFIX : 0u))); L.read_mapq.push_back((uint8_t)mapq);
    L.read_cb.push_back(cb_id);
    // CIGAR walk (htslib resolve_cigar2 semantics)
    int64_t x = pos; uint32_t y = 0;
    int64_t last_pos = -2;
    auto emit = [&](int64_t refpos, uint32_t sym, uint32_t q) {
        if (refpos != last_pos + 1 || L.seg_read.empty() || L.seg_read.back() != r) {
            L.seg_read.push_back(r); L.seg_start.push_back((int32_t)refpos); L.seg_len.push_back(0); L.seg_ev_off.push_back((int64_t)L.events.size());
        }
        ++L.seg_len.back();
        L.events.push_back(LSG_EVENT(sym, q));
        last_pos = refpos;
    };
    auto qual_at = [&](uint32_t q) -> uint32_t { return q < l_seq ? qual[q] : 0u; };
    auto base_sym = [&](uint32_t q) -> uint32_t { return q < l_seq ? NT16_SYM[(seq[q >> 1] >> ((~q & 1) << 2)) & 0xf] : 6u; };   // beyond l_qseq pysam prints 'N'
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t c = rd32(cigar + 4ull * k), op = c & 0xf, len_op = c >> 4;
        if (op == 1 || op == 4) { y += len_op; continue; }                       // I, S consume the query only
        if (!is_ref_op(op)) continue;                                             // H, P
        // indel flag of the LAST reference position of this op (peek at the next operations)
        uint32_t over = 15;                                                       // 15 = none, 4 = I, 5 = D
        if (k + 1 < n_cigar) {
            const uint32_t op2 = rd32(cigar + 4ull * (k + 1)) & 0xf;
            if (op2 == 2 && (op != 2 || g_legacy_del_merge.load(std::memory_order_relaxed))) over = 5;
            else if (op2 == 1) over = 4;
            else if (op2 == 6 && k + 2 < n_cigar) {
                uint32_t l3 = 0;
                for (uint32_t j = k + 2; j < n_cigar; ++j) {
                    const uint32_t cj = rd32(cigar + 4ull * j), oj = cj & 0xf;
                    if (oj == 1) l3 += cj >> 4;
                    else if (oj == 2 || oj == 0 || oj == 3 || oj == 7 || oj == 8) break;
                }
                if (l3 > 0) over = 4;
            }
        }
        if (op == 0 || op == 7 || op == 8) {
            for (uint32_t i = 0; i < len_op; ++i) {
                uint32_t sym = base_sym(y + i);
                if (i + 1 == len_op && over != 15) sym = over;
                emit(x + i, sym, qual_at(y + i));
            }
            x += len_op; y += len_op;
        } else if (op == 2) {                                                     // deletion: '*' -> O, quality of the next query base
            for (uint32_t i = 0; i < len_op; ++i) {
                uint32_t sym = 7;
                if (i + 1 == len_op && over != 15) sym = over;
                emit(x + i, sym, qual_at(y));
            }
            x += len_op;
        } else {                                                                  // N: '>' '<' are NA, except an indel flag on its last column
            if (len_op > 0 && over != 15) emit(x + len_op - 1, over, qual_at(y));
            x += len_op;
        }
    }
}
template <class T> T* dup_vec(const std::vector<T>& v) {
    T* p = (T*)malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (p && !v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

} // namespace

extern "C" {

typedef struct {
    int64_t n_reads, n_segs, n_events;
    int32_t* read_tid; int32_t* read_pos; uint16_t* read_flag; uint8_t* read_mapq; int32_t* read_cb;
    uint32_t* seg_read; int32_t* seg_start; int32_t* seg_len; int64_t* seg_ev_off; uint16_t* events;
    int32_t n_contigs; char* contig_names;   /* '\n'-joined */ int64_t* contig_len;
    int64_t total_reads, pass_reads, cb_not_found, cb_not_matched, mapq_filtered;
    int32_t n_barcodes; char* barcodes;      /* auto-barcode mode: the distinct cleaned CBs found, '\n'-joined, id = order of first appearance */
    int64_t n_tally; int64_t* cb_pass; int64_t* cb_low;   /* per dense barcode id (listed-barcode mode): matched reads with MAPQ >= / < min_mapq */
} lsio_decoded;

const char* lsio_last_error(void) { return g_err; }
void lsio_set_legacy_del_merge(int on) { g_legacy_del_merge.store(on ? 1 : 0); }
int lsio_get_legacy_del_merge(void) { return g_legacy_del_merge.load(); }
void lsio_set_keep_unlisted(int on) { g_keep_unlisted.store(on ? 1 : 0); }
int lsio_get_keep_unlisted(void) { return g_keep_unlisted.load(); }

void lsio_free_decoded(lsio_decoded* d) {
    if (!d) return;
    free(d->read_tid); free(d->read_pos); free(d->read_flag); free(d->read_mapq); free(d->read_cb);
    free(d->seg_read); free(d->seg_start); free(d->seg_len); free(d->seg_ev_off); free(d->events);
    free(d->contig_names); free(d->contig_len); free(d->barcodes); free(d->cb_pass); free(d->cb_low);
    free(d);
}

// ---- streaming BAM reader ----------------------------------------------------------------------------------------------------
// The compressed file is mapped, not copied; BGZF blocks are inflated a batch at a time (in parallel), records are cut out of the
// batch and the bytes of a record that continues in the next batch are carried over.  Every length field read from the file is
// checked against the bytes that are there before it is used (block sizes, header fields, record sizes, the fields inside a
// record): a truncated or corrupt file ends in an error message, never in a read past a buffer.
} // extern "C"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct lsio_stream {
    std::string path;
    const uint8_t* file = nullptr; size_t fsize = 0; int fd = -1;
    size_t off = 0;                       // next BGZF block
    std::vector<uint8_t> tail;            // bytes of an incomplete record (or of the incomplete header) from the batches so far
    bool header_done = false, eof = false;
    std::string names; std::vector<int64_t> lens; uint32_t n_ref = 0;
    std::vector<uint8_t> header_bytes;    // BAM magic .. end of the reference table (what a "wb" copy of the header writes)
    std::unordered_map<std::string, int32_t> cbmap;
    bool auto_barcodes = false; std::string auto_joined; int32_t auto_n = 0; int64_t n_tally = 0;
    int min_mapq = 0, n_threads = 1;
    ~lsio_stream() { if (file && file != (const uint8_t*)MAP_FAILED && fsize) munmap((void*)file, fsize); if (fd >= 0) close(fd); }
};

namespace {
// next batch of blocks: appends their uncompressed bytes to `data` (which starts with st.tail).  Returns -1 on a corrupt file.
int inflate_batch(lsio_stream& st, size_t max_ubytes, std::vector<uint8_t>& data) {
    std::vector<Block> blocks;
    size_t utotal = 0;
    const size_t base = st.tail.size();
    while (st.off < st.fsize && (utotal < max_ubytes || blocks.empty())) {
        if (st.off + 18 > st.fsize) { set_err("%s: truncated BGZF block header at offset %zu", st.path.c_str(), st.off); return -1; }
        const uint8_t* h = st.file + st.off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { set_err("%s is not BGZF (offset %zu)", st.path.c_str(), st.off); return -1; }
        const uint32_t xlen = rd16(h + 10);
        if (st.off + 12 + (size_t)xlen > st.fsize) { set_err("%s: truncated BGZF extra field at offset %zu", st.path.c_str(), st.off); return -1; }
        uint32_t bsize = 0; bool found = false;
        for (uint32_t q = 0; q + 4 <= xlen;) {
            const uint8_t* sf = h + 12 + q; const uint32_t slen = rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && q + 6 <= xlen) { bsize = rd16(sf + 4) + 1u; found = true; }
            q += 4 + slen;
        }
        if (!found || bsize < xlen + 20u || st.off + bsize > st.fsize) { set_err("%s: corrupt BGZF block at offset %zu", st.path.c_str(), st.off); return -1; }
        const uint32_t usize = rd32(h + bsize - 4);
        if (usize > 65536u) { set_err("%s: BGZF block at offset %zu claims %u uncompressed bytes", st.path.c_str(), st.off, usize); return -1; }
        blocks.push_back(Block{st.off + 12 + xlen, bsize - xlen - 20, usize, base + utotal});
        utotal += usize; st.off += bsize;
    }
    if (st.off >= st.fsize) st.eof = true;
    data.resize(base + utotal + 8);
    if (base) memcpy(data.data(), st.tail.data(), base);
    st.tail.clear();
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    auto worker = [&]() {
        z_stream zs;
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= blocks.size()) break;
            if (blocks[b].usize == 0) continue;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; break; }
            zs.next_in = (Bytef*)(st.file + blocks[b].off); zs.avail_in = blocks[b].csize;
            zs.next_out = data.data() + blocks[b].uoff; zs.avail_out = blocks[b].usize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; break; }
            // the block's CRC32 (RFC 1952 trailer: CRC32, ISIZE behind the deflate stream), as htslib's bgzf reader checks it: a payload that
            // still inflates to ISIZE bytes but to other bytes is a corrupt file (pysam refuses it: SplitBamCellTypes.py:51,65)
            const uint32_t want = rd32(st.file + blocks[b].off + blocks[b].csize);
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), data.data() + blocks[b].uoff, blocks[b].usize) != want) { bad = 2; break; }
        }
    };
    const int T = (int)std::min<size_t>((size_t)st.n_threads, std::max<size_t>(1, blocks.size()));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
    if (bad) { set_err(bad == 2 ? "CRC32 mismatch in a BGZF block of %s" : "inflate failed in %s", st.path.c_str()); return -1; }
    data.resize(base + utotal);
    return 0;
}

// parse the BAM header out of `data` starting at 0; returns bytes consumed, 0 when more bytes are needed, -1 on a corrupt header
int64_t parse_header(lsio_stream& st, const std::vector<uint8_t>& data) {
    const uint8_t* d = data.data(); const size_t n = data.size();
    if (n < 12) return 0;
    if (memcmp(d, "BAM\1", 4) != 0) { set_err("%s has no BAM magic", st.path.c_str()); return -1; }
    const uint64_t l_text = rd32(d + 4);
    uint64_t p = 8 + l_text;
    if (p + 4 > n) return 0;
    const uint32_t n_ref = rd32(d + p); p += 4;
    std::string names; std::vector<int64_t> lens;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (p + 4 > n) return 0;
        const uint64_t l_name = rd32(d + p); p += 4;
        if (l_name > (1u << 20)) { set_err("%s: reference name of %llu bytes in the BAM header", st.path.c_str(), (unsigned long long)l_name); return -1; }
        if (p + l_name + 4 > n) return 0;
        names.append((const char*)d + p, l_name ? strnlen((const char*)d + p, l_name - 1) : 0); names.push_back('\n'); p += l_name;
        lens.push_back((int64_t)rd32(d + p)); p += 4;
    }
    st.names = names; st.lens = lens; st.n_ref = n_ref; st.header_done = true;
    st.header_bytes.assign(d, d + p);
    return (int64_t)p;
}

// one batch -> record offsets (into data) of the complete records; the rest goes to st.tail.  Returns -1 on error, 0 at EOF with nothing left.
int next_records(lsio_stream& st, size_t max_ubytes, std::vector<uint8_t>& data, std::vector<size_t>& recs) {
    recs.clear();
    for (;;) {
        if (st.eof && st.tail.empty()) return 0;
        const size_t before = st.tail.size();
        if (!st.eof) { if (inflate_batch(st, max_ubytes, data) != 0) return -1; }
        else { data.assign(st.tail.begin(), st.tail.end()); st.tail.clear(); }
        size_t p = 0;
        if (!st.header_done) {
            const int64_t used = parse_header(st, data);
            if (used < 0) return -1;
            if (used == 0) {
                if (st.eof) { set_err("%s: truncated BAM header", st.path.c_str()); return -1; }
                st.tail.assign(data.begin(), data.end()); continue;
            }
            p = (size_t)used;
        }
        const uint8_t* d = data.data(); const size_t n = data.size();
        while (p + 4 <= n) {
            const uint64_t bs = rd32(d + p);
            if (bs < 32 || bs > (1u << 30)) { set_err("%s: record with block_size %llu", st.path.c_str(), (unsigned long long)bs); return -1; }
            if (p + 4 + bs > n) break;
            if (!record_ok(d + p + 4, (uint32_t)bs)) { set_err("%s: record whose fields exceed its block_size %llu", st.path.c_str(), (unsigned long long)bs); return -1; }
            if (rdi32(d + p + 4) >= (int32_t)st.n_ref) { set_err("%s: record on reference %d of %u", st.path.c_str(), rdi32(d + p + 4), st.n_ref); return -1; }
            {   // the CIGAR must describe the stored sequence and stay on its reference (a damaged length would otherwise be expanded
                // into up to 2^28 events per operation)
                const uint8_t* rec = d + p + 4;
                const int32_t tid = rdi32(rec), pos = rdi32(rec + 4);
                const uint32_t n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16), flag = rd16(rec + 14);
                const uint8_t* cg = rec + 32 + rec[8];
                uint64_t qlen = 0, rlen = 0;
                for (uint32_t k = 0; k < n_cigar; ++k) {
                    const uint32_t c = rd32(cg + 4ull * k), op = c & 0xf, l = c >> 4;
                    if (op > 8) { set_err("%s: CIGAR operation %u", st.path.c_str(), op); return -1; }
                    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += l;
                    if (is_ref_op(op)) rlen += l;
                }
                if (tid >= 0 && n_cigar && !(flag & 0x4)) {
                    if (l_seq && qlen != l_seq) { set_err("%s: CIGAR covers %llu query bases, the record stores %u", st.path.c_str(), (unsigned long long)qlen, l_seq); return -1; }
                    if (pos < 0 || (uint64_t)pos + rlen > (uint64_t)st.lens[(size_t)tid]) { set_err("%s: alignment at %d + %llu leaves reference %d (%lld bp)", st.path.c_str(), pos, (unsigned long long)rlen, tid, (long long)st.lens[(size_t)tid]); return -1; }
                }
            }
            recs.push_back(p); p += 4 + (size_t)bs;
        }
        if (p < n) {
            if (st.eof) { set_err("%s: truncated record at the end of the file", st.path.c_str()); return -1; }
            st.tail.assign(d + p, d + n);
        }
        if (!recs.empty() || (st.eof && st.tail.empty())) return recs.empty() ? 0 : 1;
        if (st.eof && st.tail.size() == before) { set_err("%s: truncated record at the end of the file", st.path.c_str()); return -1; }
    }
}

lsio_decoded* decode_records(lsio_stream& st, const uint8_t* d, const std::vector<size_t>& recs) {
    if (st.auto_barcodes) {
        for (size_t i = 0; i < recs.size(); ++i) {
            const char* cb; size_t cl;
            if (rdi32(d + recs[i] + 4) < 0 || !find_cb(d + recs[i] + 4, rd32(d + recs[i]), &cb, &cl)) continue;
            auto ins = st.cbmap.emplace(std::string(cb, cl), st.auto_n);
            if (ins.second) { st.auto_joined.append(cb, cl); st.auto_joined.push_back('\n'); ++st.auto_n; }
        }
    }
    const int T = (int)std::min<size_t>((size_t)st.n_threads, std::max<size_t>(1, recs.size() / 1024));
    std::vector<Local> loc((size_t)T);
    const int64_t n_tally = st.n_tally;
    if (n_tally > 0) for (auto& l : loc) { l.cb_pass.assign((size_t)n_tally, 0); l.cb_low.assign((size_t)n_tally, 0); }
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                const size_t a = recs.size() * (size_t)t / (size_t)T, b = recs.size() * (size_t)(t + 1) / (size_t)T;
                for (size_t i = a; i < b; ++i) decode_record(d + recs[i] + 4, rd32(d + recs[i]), st.cbmap, st.min_mapq, loc[(size_t)t]);
            });
        for (auto& t : th) t.join();
    }
    lsio_decoded* o = (lsio_decoded*)calloc(1, sizeof(lsio_decoded));
    int64_t R = 0, S = 0, E = 0;
    for (auto& l : loc) { R += (int64_t)l.read_tid.size(); S += (int64_t)l.seg_read.size(); E += (int64_t)l.events.size(); }
    o->n_reads = R; o->n_segs = S; o->n_events = E;
    auto al = [](size_t n, size_t sz) { return malloc(sz * (n ? n : 1)); };
    o->read_tid = (int32_t*)al(R, 4); o->read_pos = (int32_t*)al(R, 4); o->read_flag = (uint16_t*)al(R, 2); o->read_mapq = (uint8_t*)al(R, 1);
    o->read_cb = (int32_t*)al(R, 4); o->seg_read = (uint32_t*)al(S, 4); o->seg_start = (int32_t*)al(S, 4); o->seg_len = (int32_t*)al(S, 4);
    o->seg_ev_off = (int64_t*)al(S, 8); o->events = (uint16_t*)al(E, 2);
    // every thread copies its own part to its place in the merged arrays (first touch of the big buffers in parallel too)
    std::vector<int64_t> r_at(loc.size() + 1, 0), s_at(loc.size() + 1, 0), e_at(loc.size() + 1, 0);
    for (size_t q = 0; q < loc.size(); ++q) {
        r_at[q + 1] = r_at[q] + (int64_t)loc[q].read_tid.size(); s_at[q + 1] = s_at[q] + (int64_t)loc[q].seg_read.size();
        e_at[q + 1] = e_at[q] + (int64_t)loc[q].events.size();
    }
    {
        std::vector<std::thread> th;
        for (size_t q = 0; q < loc.size(); ++q)
            th.emplace_back([&, q]() {
                const Local& l = loc[q];
                const int64_t r0 = r_at[q], s0 = s_at[q], e0 = e_at[q];
                const size_t nr = l.read_tid.size(), ns = l.seg_read.size(), ne = l.events.size();
                if (nr) { memcpy(o->read_tid + r0, l.read_tid.data(), nr * 4); memcpy(o->read_pos + r0, l.read_pos.data(), nr * 4);
                          memcpy(o->read_flag + r0, l.read_flag.data(), nr * 2); memcpy(o->read_mapq + r0, l.read_mapq.data(), nr);
                          memcpy(o->read_cb + r0, l.read_cb.data(), nr * 4); }
                for (size_t i = 0; i < ns; ++i) {
                    o->seg_read[s0 + i] = l.seg_read[i] + (uint32_t)r0; o->seg_start[s0 + i] = l.seg_start[i]; o->seg_len[s0 + i] = l.seg_len[i];
                    o->seg_ev_off[s0 + i] = l.seg_ev_off[i] + e0;
                }
                if (ne) memcpy(o->events + e0, l.events.data(), ne * 2);
            });
        for (auto& t : th) t.join();
    }
    for (auto& l : loc) {
        o->total_reads += l.total; o->pass_reads += l.pass; o->cb_not_found += l.cb_not_found; o->cb_not_matched += l.cb_not_matched;
        o->mapq_filtered += l.mapq;
    }
    o->n_tally = n_tally;
    o->cb_pass = (int64_t*)calloc((size_t)(n_tally ? n_tally : 1), 8); o->cb_low = (int64_t*)calloc((size_t)(n_tally ? n_tally : 1), 8);
    for (auto& l : loc) for (size_t i = 0; i < l.cb_pass.size(); ++i) { o->cb_pass[i] += l.cb_pass[i]; o->cb_low[i] += l.cb_low[i]; }
    o->n_contigs = (int32_t)st.n_ref;
    o->contig_names = (char*)malloc(st.names.size() + 1); memcpy(o->contig_names, st.names.c_str(), st.names.size() + 1);
    o->contig_len = dup_vec(st.lens);
    o->n_barcodes = st.auto_n;
    o->barcodes = (char*)malloc(st.auto_joined.size() + 1); memcpy(o->barcodes, st.auto_joined.c_str(), st.auto_joined.size() + 1);
    return o;
}
} // namespace

extern "C" {

// barcodes: n_barcodes cleaned barcode strings joined by '\n' (n_barcodes < 0: every distinct cleaned CB of the file is a cell, the
// way BaseCellCounter sees a per-cell-type BAM); ids[i] = dense id of barcode i (NULL: i).
int lsio_stream_open(const char* path, const char* barcodes, int32_t n_barcodes, const int32_t* ids, int32_t min_mapq, int32_t n_threads,
                     lsio_stream** out) {
    if (!path || !out) { set_err("lsio_stream_open: bad arguments"); return -2; }
    *out = nullptr;
    lsio_stream* st = new lsio_stream();
    st->path = path; st->min_mapq = min_mapq;
    st->n_threads = n_threads > 0 ? n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
    st->cbmap.reserve((size_t)(n_barcodes > 0 ? n_barcodes : 0) * 2 + 16);
    const char* s = barcodes ? barcodes : "";
    for (int32_t i = 0; i < n_barcodes; ++i) {
        const char* e = strchr(s, '\n'); size_t l = e ? (size_t)(e - s) : strlen(s);
        st->cbmap[std::string(s, l)] = ids ? ids[i] : i;                          // duplicates: last wins (to_dict, :31)
        s += l + (e ? 1 : 0);
    }
    st->auto_barcodes = n_barcodes < 0;
    if (n_barcodes > 0) for (auto& kv : st->cbmap) st->n_tally = kv.second + 1 > st->n_tally ? kv.second + 1 : st->n_tally;
    st->fd = open(path, O_RDONLY);
    struct stat sb;
    if (st->fd < 0 || fstat(st->fd, &sb) != 0) { set_err("cannot open %s", path); delete st; return -1; }
    st->fsize = (size_t)sb.st_size;
    if (st->fsize) {
        st->file = (const uint8_t*)mmap(nullptr, st->fsize, PROT_READ, MAP_PRIVATE, st->fd, 0);
        if (st->file == (const uint8_t*)MAP_FAILED) { st->file = nullptr; set_err("cannot map %s", path); delete st; return -1; }
        (void)madvise((void*)st->file, st->fsize, MADV_SEQUENTIAL);
    }
    *out = st;
    return 0;
}

void lsio_stream_close(lsio_stream* st) { delete st; }

// Decodes the next batch (about max_ubytes of uncompressed BAM, at least one record; <= 0: everything that is left).  Returns 1 with
// *out set, 0 at the end of the file (*out = NULL), < 0 on error.  The counters of the batch are in *out; the barcode list of the
// auto-barcode mode grows from batch to batch (ids stay stable).
int lsio_stream_next(lsio_stream* st, int64_t max_ubytes, lsio_decoded** out) {
    if (!st || !out) { set_err("lsio_stream_next: bad arguments"); return -2; }
    *out = nullptr;
    std::vector<uint8_t> data; std::vector<size_t> recs;
    const int rc = next_records(*st, max_ubytes > 0 ? (size_t)max_ubytes : (size_t)-1, data, recs);
    if (rc < 0) return -1;
    if (rc == 0) {
        if (!st->header_done) { set_err("%s: no BAM header", st->path.c_str()); return -1; }
        return 0;
    }
    *out = decode_records(*st, data.data(), recs);
    return 1;
}

int lsio_decode_bam(const char* path, const char* barcodes, int32_t n_barcodes, const int32_t* ids, int32_t min_mapq, int32_t n_threads,
                    lsio_decoded** out) {
    if (!path || !out) { set_err("lsio_decode_bam: bad arguments"); return -2; }
    *out = nullptr;
    lsio_stream* st = nullptr;
    int rc = lsio_stream_open(path, barcodes, n_barcodes, ids, min_mapq, n_threads, &st);
    if (rc != 0) return rc;
    std::vector<uint8_t> data; std::vector<size_t> recs;
    rc = next_records(*st, (size_t)-1, data, recs);
    if (rc < 0 || !st->header_done) { if (rc >= 0) set_err("%s: no BAM header", path); delete st; return -1; }
    *out = decode_records(*st, data.data(), recs);
    delete st;
    return 0;
}

// ---- BGZF / BAM writing -------------------------------------------------------------------------
struct BgzfWriter {
    FILE* f = nullptr; std::vector<uint8_t> buf; bool ok = true;
    void flush_block(const uint8_t* src, size_t n) {
        uint8_t out[70000];
        z_stream zs; memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n; zs.next_out = out + 18; zs.avail_out = sizeof(out) - 26;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) ok = false;
        const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
        const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
        memcpy(out, hdr, 16);
        const uint32_t bsize = clen + 25;
        out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
        uint8_t* t = out + 18 + clen;
        for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
        if (fwrite(out, 1, 18 + clen + 8, f) != 18 + clen + 8) ok = false;
    }
    void write(const void* p, size_t n) {
        const uint8_t* s = (const uint8_t*)p;
        while (n) {
            const size_t room = 0xff00 - buf.size(), take = n < room ? n : room;
            buf.insert(buf.end(), s, s + take); s += take; n -= take;
            if (buf.size() >= 0xff00) { flush_block(buf.data(), buf.size()); buf.clear(); }
        }
    }
    // one BAM record, the way htslib's bam_write1 does it (bgzf_flush_try): a record that does not fit the open block starts a new
    // one, so that every block begins on a record boundary unless a record is longer than a block
    void write_record(const void* p, size_t n) {
        if (!buf.empty() && buf.size() + n > 0xff00) { flush_block(buf.data(), buf.size()); buf.clear(); }
        write(p, n);
    }
    void close() {
        if (!buf.empty()) { flush_block(buf.data(), buf.size()); buf.clear(); }
        flush_block(nullptr, 0);                                                 // EOF marker block
        if (f) fclose(f);
        f = nullptr;
    }
};

static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }

// ---- the synthetic BAM, written on every host thread -------------------------------------------------------------------------------
// The file is the one a single BgzfWriter fed record by record would write (same block cuts: a record that does not fit the open block
// starts a new one); the work is spread: read headers, the records of a round, the deflate of its blocks and the FASTA lines are
// each done by all threads, the file is written in order.  (The C2 BAM of 10 M reads is 9.9 GB: one thread took 7 minutes.)
extern "C++" {
static int synth_threads() { const unsigned h = std::thread::hardware_concurrency(); return (int)std::min(64u, std::max(1u, h)); }

template <class F> static void synth_parallel(int64_t n_items, int64_t grain, F&& fn) {   // fn(lo, hi) over [0, n_items) in grains
    const int64_t n_chunks = (n_items + grain - 1) / grain;
    const int T = (int)std::min<int64_t>(synth_threads(), n_chunks);
    if (T <= 1) { if (n_items > 0) fn((int64_t)0, n_items); return; }
    std::atomic<int64_t> next(0);
    auto worker = [&]() { for (;;) { const int64_t c = next.fetch_add(1); if (c >= n_chunks) return; fn(c * grain, std::min(n_items, (c + 1) * grain)); } };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

static void bgzf_block(const uint8_t* src, size_t n, uint8_t* out /* >= 70000 */, uint32_t* out_len, bool* ok) {
    z_stream zs; memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef*)src; zs.avail_in = (uInt)n; zs.next_out = out + 18; zs.avail_out = 70000 - 26;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) *ok = false;
    const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
    const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    const uint32_t bsize = clen + 25;
    out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
    uint8_t* t = out + 18 + clen;
    for (int i = 0; i < 4; ++i) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
    *out_len = 18 + clen + 8;
}

struct SynthScratch { std::vector<uint32_t> cig; std::string seq; std::vector<uint8_t> qual; };

// one read of the model as a BAM record appended to `rec` (CB:Z tag, soft clips, indels, introns)
static void synth_record(const lsg_synth_model* m, int64_t ig, const sm_read& r, int32_t pos, const char* barcode_suffix, SynthScratch& sc,
                         std::vector<uint8_t>& rec) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    auto code_of = [](char c) -> uint8_t { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; };
    std::vector<uint32_t>& cig = sc.cig; std::string& seq = sc.seq; std::vector<uint8_t>& qual = sc.qual;
    cig.clear(); seq.clear(); qual.clear();
    auto push_op = [&](uint32_t op, uint32_t len) { if (!len) return; if (!cig.empty() && (cig.back() & 0xf) == op) cig.back() += len << 4; else cig.push_back((len << 4) | op); };
    for (int32_t c = 0; c < r.clip5; ++c) { const uint64_t h = sm_hash(m->seed, (uint64_t)ig, (uint64_t)(100000 + c), SM_D_INS); seq.push_back(acgt[h & 3]); qual.push_back((uint8_t)(2 + (h >> 8) % 39)); }
    push_op(4, (uint32_t)r.clip5);
    const int32_t end = r.t_off + r.t_len;
    for (int32_t x = r.e0; x <= r.e1; ++x) {
        const int32_t xt0 = m->exon_cum[x], xt1 = xt0 + m->exon_len[x];
        const int32_t lo = r.t_off > xt0 ? r.t_off : xt0, hi = end < xt1 ? end : xt1;
        if (x > r.e0) push_op(3, (uint32_t)(m->exon_start[x] - (m->exon_start[x - 1] + m->exon_len[x - 1])));
        for (int32_t j = lo; j < hi; ++j) {
            const int32_t blk = j >> 3, k = j & 7;
            const int32_t ind = sm_block_indel(m, ig, blk, lo, hi);
            if (ind > 0 && k >= 2 && k <= 1 + ind) { push_op(2, 1); continue; }      // deleted base
            const uint32_t s = sm_base_call(m, ig, &r, j, (int64_t)m->exon_start[x] + (j - xt0));
            seq.push_back((char)sm_base_of_sym(s)); qual.push_back((uint8_t)sm_qual(m, ig, j));
            push_op(0, 1);
            if (ind < 0 && k == 3) {
                for (int32_t c = 0; c < -ind; ++c) { const uint64_t h = sm_hash(m->seed, (uint64_t)ig, (uint64_t)((uint32_t)j * 4u + (uint32_t)c), SM_D_INS); seq.push_back(acgt[h & 3]); qual.push_back((uint8_t)(2 + (h >> 8) % 39)); }
                push_op(1, (uint32_t)(-ind));
            }
        }
    }
    for (int32_t c = 0; c < r.clip3; ++c) { const uint64_t h = sm_hash(m->seed, (uint64_t)ig, (uint64_t)(200000 + c), SM_D_INS); seq.push_back(acgt[h & 3]); qual.push_back((uint8_t)(2 + (h >> 8) % 39)); }
    push_op(4, (uint32_t)r.clip3);
    char name[32]; const int ln = snprintf(name, sizeof(name), "r%lld", (long long)ig) + 1;
    const size_t at = rec.size();
    put32(rec, 0);                                                           // block_size placeholder
    put32(rec, (uint32_t)r.tid); put32(rec, (uint32_t)pos);
    rec.push_back((uint8_t)ln); rec.push_back(r.mapq); rec.push_back(0x48); rec.push_back(0x12);   // bin (unused)
    rec.push_back((uint8_t)(cig.size() & 0xff)); rec.push_back((uint8_t)(cig.size() >> 8));
    rec.push_back((uint8_t)(r.flag & 0xff)); rec.push_back((uint8_t)(r.flag >> 8));
    put32(rec, (uint32_t)seq.size()); put32(rec, 0xffffffffu); put32(rec, 0xffffffffu); put32(rec, 0);
    rec.insert(rec.end(), name, name + ln);
    for (uint32_t c : cig) put32(rec, c);
    for (size_t q = 0; q < seq.size(); q += 2) rec.push_back((uint8_t)((code_of(seq[q]) << 4) | (q + 1 < seq.size() ? code_of(seq[q + 1]) : 0)));
    rec.insert(rec.end(), qual.begin(), qual.end());
    const uint8_t nh[7] = {'N', 'H', 'C', 1, 0, 0, 0}; rec.insert(rec.end(), nh, nh + 4);
    if (r.cb != -1) {
        char bc[17]; sm_barcode(m->seed, r.cb >= 0 ? r.cb : ig, r.cb < 0, bc);
        rec.push_back('C'); rec.push_back('B'); rec.push_back('Z'); rec.insert(rec.end(), bc, bc + 16);
        if (barcode_suffix) rec.insert(rec.end(), barcode_suffix, barcode_suffix + strlen(barcode_suffix));
        rec.push_back(0);
    }
    const uint32_t bs = (uint32_t)(rec.size() - at) - 4;
    for (int b = 0; b < 4; ++b) rec[at + (size_t)b] = (uint8_t)(bs >> (8 * b));
}

}  // extern "C++"

// Writes the model's reads as a coordinate-sorted BAM (CB:Z tags, soft clips, indels, introns) and,
// optionally, the reference FASTA.  barcode_suffix (e.g. "-1") is appended to every CB value.
int lsio_synth_bam(const lsg_synth_model* m, const char* contig_names /* '\n'-joined */, const int64_t* contig_len, const char* bam_path,
                   const char* fasta_path, const char* barcode_suffix) {
    if (!m || !bam_path) { set_err("lsio_synth_bam: bad arguments"); return -2; }
    std::vector<std::string> names;
    { const char* s = contig_names; for (int i = 0; i < m->n_contigs; ++i) { const char* e = strchr(s, '\n'); size_t l = e ? (size_t)(e - s) : strlen(s); names.emplace_back(s, l); s += l + (e ? 1 : 0); } }
    if (fasta_path && *fasta_path) {
        FILE* ff = fopen(fasta_path, "w");
        if (!ff) { set_err("lsio_synth_bam: cannot write %s", fasta_path); return -1; }
        std::vector<char> text;
        for (int t = 0; t < m->n_contigs; ++t) {
            fprintf(ff, ">%s\n", names[(size_t)t].c_str());
            const int64_t len = contig_len[t], n_lines = (len + 59) / 60;
            text.resize((size_t)(len + n_lines));
            synth_parallel(n_lines, 1 << 14, [&](int64_t l0, int64_t l1) {      // line l: bases [60 l, 60 l + 60) at text offset 61 l
                for (int64_t l = l0; l < l1; ++l) {
                    char* o = text.data() + 61 * l;
                    const int64_t p1 = std::min(len, 60 * l + 60);
                    for (int64_t p = 60 * l; p < p1; ++p) *o++ = (char)sm_ref_base(m->seed, t, p);
                    *o = '\n';
                }
            });
            if (fwrite(text.data(), 1, text.size(), ff) != text.size()) { fclose(ff); set_err("lsio_synth_bam: write failed (%s)", fasta_path); return -1; }
        }
        fclose(ff);
    }
    const int64_t R = m->n_reads;
    std::vector<sm_read> hdr((size_t)R);
    std::vector<int32_t> pos((size_t)R);
    struct Key { uint64_t k; int64_t i; };
    std::vector<Key> order((size_t)R);
    synth_parallel(R, 1 << 14, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            sm_read_header(m, i + m->read_base, &hdr[(size_t)i]);
            const sm_read& r = hdr[(size_t)i];
            pos[(size_t)i] = m->exon_start[r.e0] + (r.t_off - m->exon_cum[r.e0]);
            order[(size_t)i] = Key{((uint64_t)(uint32_t)r.tid << 32) | (uint32_t)pos[(size_t)i], i};
        }
    });
    std::sort(order.begin(), order.end(), [](const Key& a, const Key& b) { return a.k != b.k ? a.k < b.k : a.i < b.i; });   // (tid, pos), ties in read order
    FILE* f = fopen(bam_path, "wb");
    if (!f) { set_err("lsio_synth_bam: cannot write %s", bam_path); return -1; }
    std::vector<uint8_t> pending;                                                // the bytes not yet in a closed block: starts with the open block
    {
        std::vector<uint8_t>& h = pending; h.insert(h.end(), {'B', 'A', 'M', 1});
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int t = 0; t < m->n_contigs; ++t) text += "@SQ\tSN:" + names[(size_t)t] + "\tLN:" + std::to_string(contig_len[t]) + "\n";
        put32(h, (uint32_t)text.size()); h.insert(h.end(), text.begin(), text.end());
        put32(h, (uint32_t)m->n_contigs);
        for (int t = 0; t < m->n_contigs; ++t) { put32(h, (uint32_t)names[(size_t)t].size() + 1); h.insert(h.end(), names[(size_t)t].begin(), names[(size_t)t].end()); h.push_back(0); put32(h, (uint32_t)contig_len[t]); }
    }
    // block cuts of a stream of writes, as BgzfWriter makes them: `open` bytes are in the open block, which starts at `start`
    std::vector<std::pair<size_t, uint32_t>> blocks;                             // (offset in pending, bytes)
    size_t start = 0, open = 0;
    auto close_block = [&]() { blocks.emplace_back(start, (uint32_t)open); start += open; open = 0; };
    auto cut_write = [&](size_t n) { while (n) { const size_t room = 0xff00 - open, take = n < room ? n : room; open += take; n -= take; if (open >= 0xff00) close_block(); } };
    auto cut_record = [&](size_t n) { if (open && open + n > 0xff00) close_block(); cut_write(n); };
    cut_write(pending.size());
    bool ok = true;
    const int64_t BATCH = 512, ROUND = std::max<int64_t>(65536, (int64_t)synth_threads() * 2048);
    std::vector<std::vector<uint8_t>> bb; std::vector<std::vector<uint32_t>> bs;
    std::vector<uint8_t> zout; std::vector<uint32_t> zlen; std::vector<uint8_t> okv;
    for (int64_t r0 = 0; r0 < R || r0 == 0; r0 += ROUND) {
        const int64_t r1 = std::min(R, r0 + ROUND), nb = (r1 - r0 + BATCH - 1) / BATCH;
        bb.assign((size_t)nb, {}); bs.assign((size_t)nb, {});
        synth_parallel(nb, 1, [&](int64_t b0, int64_t b1) {
            SynthScratch sc;
            for (int64_t b = b0; b < b1; ++b) {
                std::vector<uint8_t>& out = bb[(size_t)b]; std::vector<uint32_t>& sz = bs[(size_t)b];
                for (int64_t oi = r0 + b * BATCH; oi < std::min(r1, r0 + (b + 1) * BATCH); ++oi) {
                    const int64_t i = order[(size_t)oi].i; const size_t at = out.size();
                    synth_record(m, i + m->read_base, hdr[(size_t)i], pos[(size_t)i], barcode_suffix, sc, out);
                    sz.push_back((uint32_t)(out.size() - at));
                }
            }
        });
        std::vector<size_t> at((size_t)nb + 1, pending.size());
        for (int64_t b = 0; b < nb; ++b) at[(size_t)b + 1] = at[(size_t)b] + bb[(size_t)b].size();
        pending.resize(at[(size_t)nb]);
        synth_parallel(nb, 16, [&](int64_t b0, int64_t b1) { for (int64_t b = b0; b < b1; ++b) if (!bb[(size_t)b].empty()) memcpy(pending.data() + at[(size_t)b], bb[(size_t)b].data(), bb[(size_t)b].size()); });
        for (int64_t b = 0; b < nb; ++b) for (uint32_t n : bs[(size_t)b]) cut_record(n);
        if (r1 >= R && open) close_block();
        const int64_t nblk = (int64_t)blocks.size();
        zout.resize((size_t)nblk * 70000); zlen.assign((size_t)nblk, 0); okv.assign((size_t)nblk, 1);
        synth_parallel(nblk, 4, [&](int64_t b0, int64_t b1) {
            for (int64_t b = b0; b < b1; ++b) { bool k = true; bgzf_block(pending.data() + blocks[(size_t)b].first, blocks[(size_t)b].second, zout.data() + (size_t)b * 70000, &zlen[(size_t)b], &k); okv[(size_t)b] = k; }
        });
        for (int64_t b = 0; b < nblk; ++b) { ok = ok && okv[(size_t)b]; if (fwrite(zout.data() + (size_t)b * 70000, 1, zlen[(size_t)b], f) != zlen[(size_t)b]) ok = false; }
        blocks.clear();
        pending.erase(pending.begin(), pending.begin() + (ptrdiff_t)start);     // keep the open block
        start = 0;
        if (r1 >= R) break;
    }
    { uint8_t eof[70000]; uint32_t n = 0; bgzf_block(nullptr, 0, eof, &n, &ok); if (fwrite(eof, 1, n, f) != n) ok = false; }   // EOF marker block
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_err("lsio_synth_bam: write failed"); return -1; }
    return 0;
}

// SplitBamCellTypes' BAM outputs (split_bam, SplitBamCellTypes.py:39-192): one BAM per cell type with the
// records whose cleaned CB maps to it and whose MAPQ >= min_mapq; header copied (template=infile, :57).
// celltype_of_barcode[i] in [0, n_ct) for barcode i of the '\n'-joined list; out_paths '\n'-joined.
int lsio_split_bam(const char* path, const char* barcodes, int32_t n_barcodes, const uint8_t* celltype_of_barcode, int32_t n_ct,
                   const char* out_paths, int32_t min_mapq, int64_t* counters /* total, pass, cb_not_found, cb_not_matched, mapq */) {
    if (!path || !barcodes || !out_paths || n_ct <= 0) { set_err("lsio_split_bam: bad arguments"); return -2; }
    std::unordered_map<std::string, int32_t> ctmap;
    { const char* s = barcodes; for (int32_t i = 0; i < n_barcodes; ++i) { const char* e = strchr(s, '\n'); size_t l = e ? (size_t)(e - s) : strlen(s); ctmap[std::string(s, l)] = celltype_of_barcode[i]; s += l + (e ? 1 : 0); } }
    std::vector<std::string> outs;
    { const char* s = out_paths; for (int i = 0; i < n_ct; ++i) { const char* e = strchr(s, '\n'); size_t l = e ? (size_t)(e - s) : strlen(s); outs.emplace_back(s, l); s += l + (e ? 1 : 0); } }
    lsio_stream* st = nullptr;
    if (lsio_stream_open(path, nullptr, 0, nullptr, min_mapq, 0, &st) != 0) return -1;
    std::vector<uint8_t> data; std::vector<size_t> recs;
    const int rcs = next_records(*st, (size_t)-1, data, recs);           // validated: block sizes, header, every record's fields
    if (rcs < 0 || !st->header_done) { if (rcs >= 0) set_err("lsio_split_bam: %s has no BAM header", path); delete st; return -1; }
    const std::vector<uint8_t> header = st->header_bytes;
    delete st;
    const uint8_t* d = data.data();
    std::vector<BgzfWriter> w((size_t)n_ct);
    for (int i = 0; i < n_ct; ++i) { w[(size_t)i].f = fopen(outs[(size_t)i].c_str(), "wb"); if (!w[(size_t)i].f) { set_err("lsio_split_bam: cannot write %s", outs[(size_t)i].c_str()); return -1; } w[(size_t)i].write(header.data(), header.size()); }
    int64_t cnt[5] = {0, 0, 0, 0, 0};
    for (const size_t rec_at : recs) {
        const uint32_t bs = rd32(d + rec_at); const uint8_t* rec = d + rec_at + 4;
        if (rdi32(rec) < 0) continue;
        ++cnt[0];
        const char* cb; size_t cl;
        if (!find_cb(rec, bs, &cb, &cl)) { ++cnt[2]; continue; }
        auto it = ctmap.find(std::string(cb, cl));
        if (it == ctmap.end()) { ++cnt[3]; continue; }
        if (min_mapq > 0 && (int)rec[9] < min_mapq) { ++cnt[4]; continue; }
        ++cnt[1];
        w[(size_t)it->second].write_record(d + rec_at, 4 + bs);
    }
    bool ok = true;
    for (auto& x : w) { x.close(); ok = ok && x.ok; }
    if (counters) memcpy(counters, cnt, sizeof(cnt));
    if (!ok) { set_err("lsio_split_bam: write failed"); return -1; }
    return 0;
}

// Host evaluation of the model's read-record arrays (the same arrays lsg_synth_reads generates in HBM).
int lsio_synth_records(const lsg_synth_model* m, lsio_decoded** out) {
    if (!m || !out) { set_err("lsio_synth_records: bad arguments"); return -2; }
    Local L;
    for (int64_t i = 0; i < m->n_reads; ++i) {
        const int64_t ig = i + m->read_base;
        sm_read r; sm_read_header(m, ig, &r);
        L.read_tid.push_back(r.tid); L.read_pos.push_back(m->exon_start[r.e0] + (r.t_off - m->exon_cum[r.e0]));
        L.read_flag.push_back(r.flag); L.read_mapq.push_back(r.mapq); L.read_cb.push_back(r.cb >= 0 ? r.cb : -1);
        const int32_t end = r.t_off + r.t_len;
        const bool phased = m->layout == LSG_LAYOUT_PHASED;                  // (include/longsom_hip.h: the gaps hold 0)
        const size_t base = L.events.size();
        for (int32_t x = r.e0; x <= r.e1; ++x) {
            const int32_t xt0 = m->exon_cum[x], xt1 = xt0 + m->exon_len[x];
            const int32_t lo = r.t_off > xt0 ? r.t_off : xt0, hi = end < xt1 ? end : xt1;
            const int32_t st = m->exon_start[x] + (lo - xt0);
            if (phased) L.events.resize(base + (size_t)sm_phase_place((int64_t)(L.events.size() - base), st), 0);
            L.seg_read.push_back((uint32_t)i); L.seg_start.push_back(st); L.seg_len.push_back(hi - lo);
            L.seg_ev_off.push_back((int64_t)L.events.size());
            for (int32_t j = lo; j < hi; ++j) L.events.push_back(sm_event(m, ig, &r, j, xt0, xt1, m->exon_start[x]));
        }
        if (phased) L.events.resize(base + (((L.events.size() - base) + 127) & ~(size_t)127), 0);
    }
    lsio_decoded* o = (lsio_decoded*)calloc(1, sizeof(lsio_decoded));
    o->n_reads = (int64_t)L.read_tid.size(); o->n_segs = (int64_t)L.seg_read.size(); o->n_events = (int64_t)L.events.size();
    o->read_tid = dup_vec(L.read_tid); o->read_pos = dup_vec(L.read_pos); o->read_flag = dup_vec(L.read_flag); o->read_mapq = dup_vec(L.read_mapq);
    o->read_cb = dup_vec(L.read_cb); o->seg_read = dup_vec(L.seg_read); o->seg_start = dup_vec(L.seg_start); o->seg_len = dup_vec(L.seg_len);
    o->seg_ev_off = dup_vec(L.seg_ev_off); o->events = dup_vec(L.events);
    *out = o;
    return 0;
}

void lsio_barcode(uint64_t seed, int64_t cb, char* out17) { sm_barcode(seed, cb, 0, out17); }

// reference bases of one contig of the synthetic genome (host twin of lsg_synth_reference)
void lsio_ref_bases(uint64_t seed, int32_t tid, int64_t len, uint8_t* out) {
    for (int64_t p = 0; p < len; ++p) out[p] = sm_ref_base(seed, tid, p);
}

} // extern "C"

// ---- a BAM index with a linear index only (BAI: SAM specification section 5.2), for BAMs this repository writes ----------------------------
// The sharded run (longsom_amd/regions.py) cuts a coordinate-sorted BAM by its .bai's linear index — per reference and 16 kb window the
// smallest virtual offset (compressed block start << 16 | offset inside the block) of the alignments overlapping the window — which is
// what the reference's workers use too when they fetch their window through the index (BaseCellCounter.py:190-191; the .bai is a rule
// input, rules/SNVCalling.smk:6-7).  samtools writes that file for real data; this builder writes one for the synthetic and fixture BAMs
// (no binning index: n_bin = 0 — the reader here needs the linear index only).  Unset windows are filled as htslib fills them (with the
// window before; leading ones with the reference's first offset).
namespace {
struct BaiBuild {
    std::vector<std::vector<uint64_t>> lin;           // per reference
    void add(int32_t tid, int64_t beg, int64_t end, uint64_t voff) {
        if (tid < 0 || (size_t)tid >= lin.size()) return;
        if (end <= beg) end = beg + 1;
        auto& v = lin[(size_t)tid];
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (v.size() <= w1) v.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (v[w] == 0 || voff < v[w]) v[w] = voff;
    }
};
}
extern "C" int lsio_build_bai(const char* bam_path, const char* bai_path) {
    int fd = open(bam_path, O_RDONLY);
    if (fd < 0) { set_err("cannot open %s", bam_path); return -1; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); set_err("cannot stat %s", bam_path); return -1; }
    const size_t fsize = (size_t)sb.st_size;
    const uint8_t* file = (const uint8_t*)mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (file == (const uint8_t*)MAP_FAILED) { close(fd); set_err("cannot map %s", bam_path); return -1; }
    auto bail = [&](int rc) { munmap((void*)file, fsize); close(fd); return rc; };
    std::vector<uint8_t> buf; uint64_t buf_g0 = 0;           // unconsumed uncompressed bytes and the stream offset of the first of them
    std::vector<uint64_t> blk_ustart, blk_coff;              // per block: where it starts in the uncompressed stream / in the file
    size_t bptr = 0;
    uint64_t utotal = 0; size_t off = 0;
    bool header_done = false; uint32_t n_ref = 0;
    BaiBuild bb;
    std::vector<uint8_t> out(65536);
    auto voffset = [&](uint64_t g) {
        while (bptr + 1 < blk_ustart.size() && blk_ustart[bptr + 1] <= g) ++bptr;
        return (blk_coff[bptr] << 16) | (g - blk_ustart[bptr]);
    };
    size_t p = 0;                                            // parse position inside buf
    while (off < fsize) {
        if (off + 18 > fsize) { set_err("%s: truncated BGZF block header", bam_path); return bail(-1); }
        const uint8_t* h = file + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { set_err("%s is not BGZF", bam_path); return bail(-1); }
        const uint32_t xlen = rd16(h + 10);
        uint32_t bsize = 0; bool found = false;
        if (off + 12 + (size_t)xlen > fsize) { set_err("%s: truncated BGZF extra field", bam_path); return bail(-1); }
        for (uint32_t q = 0; q + 4 <= xlen;) {
            const uint8_t* sf = h + 12 + q; const uint32_t slen = rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && q + 6 <= xlen) { bsize = rd16(sf + 4) + 1u; found = true; }
            q += 4 + slen;
        }
        if (!found || bsize < xlen + 20u || off + bsize > fsize) { set_err("%s: corrupt BGZF block", bam_path); return bail(-1); }
        const uint32_t usize = rd32(h + bsize - 4);
        if (usize > 65536u) { set_err("%s: BGZF block of %u bytes", bam_path, usize); return bail(-1); }
        if (usize) {
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { set_err("zlib"); return bail(-1); }
            zs.next_in = (Bytef*)(h + 12 + xlen); zs.avail_in = bsize - xlen - 20; zs.next_out = out.data(); zs.avail_out = usize;
            const int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { set_err("%s: inflate failed", bam_path); return bail(-1); }
            blk_ustart.push_back(utotal); blk_coff.push_back((uint64_t)off);
            buf.insert(buf.end(), out.begin(), out.begin() + usize);
            utotal += usize;
        }
        off += bsize;
        // ---- what is complete in buf
        const uint8_t* d = buf.data(); const size_t n = buf.size();
        if (!header_done) {
            if (n < 12) continue;
            if (memcmp(d, "BAM\1", 4) != 0) { set_err("%s has no BAM magic", bam_path); return bail(-1); }
            uint64_t q = 8 + (uint64_t)rd32(d + 4);
            if (q + 4 > n) continue;
            n_ref = rd32(d + q); q += 4;
            bool complete = true;
            for (uint32_t i = 0; i < n_ref; ++i) {
                if (q + 4 > n) { complete = false; break; }
                const uint64_t l_name = rd32(d + q);
                if (q + 4 + l_name + 4 > n) { complete = false; break; }
                q += 4 + l_name + 4;
            }
            if (!complete) continue;
            bb.lin.assign(n_ref, {});
            header_done = true; p = (size_t)q;
        }
        while (p + 4 <= n) {
            const uint64_t bs = rd32(d + p);
            if (bs < 32 || bs > (1u << 30)) { set_err("%s: record with block_size %llu", bam_path, (unsigned long long)bs); return bail(-1); }
            if (p + 4 + bs > n) break;
            const uint8_t* rec = d + p + 4;
            if (!record_ok(rec, (uint32_t)bs)) { set_err("%s: record whose fields exceed its block_size", bam_path); return bail(-1); }
            const int32_t tid = rdi32(rec), pos = rdi32(rec + 4);
            const uint32_t n_cigar = rd16(rec + 12);
            const uint8_t* cg = rec + 32 + rec[8];
            int64_t rlen = 0;
            for (uint32_t k = 0; k < n_cigar; ++k) { const uint32_t c = rd32(cg + 4ull * k); if (is_ref_op(c & 0xf)) rlen += c >> 4; }
            if (tid >= 0 && pos >= 0) bb.add(tid, pos, (int64_t)pos + (rlen > 0 ? rlen : 1), voffset(buf_g0 + p));
            p += 4 + (size_t)bs;
        }
        if (p > (1u << 22)) { buf.erase(buf.begin(), buf.begin() + (long)p); buf_g0 += p; p = 0; }
    }
    if (!header_done) { set_err("%s: truncated BAM header", bam_path); return bail(-1); }
    FILE* f = fopen(bai_path, "wb");
    if (!f) { set_err("cannot write %s", bai_path); return bail(-1); }
    auto w32 = [&](uint32_t x) { fwrite(&x, 4, 1, f); };
    fwrite("BAI\1", 1, 4, f); w32(n_ref);
    for (uint32_t r = 0; r < n_ref; ++r) {
        auto& v = bb.lin[r];
        uint64_t first = 0; for (uint64_t x : v) if (x) { first = x; break; }
        for (size_t w = 0; w < v.size(); ++w) if (v[w] == 0) v[w] = w ? v[w - 1] : first;
        w32(0); w32((uint32_t)v.size());
        if (!v.empty()) fwrite(v.data(), 8, v.size(), f);
    }
    const bool ok = fclose(f) == 0;
    if (!ok) set_err("write to %s failed", bai_path);
    return bail(ok ? 0 : -1);
}
