// Device-side BAM ingest: compressed BAM bytes in, tile store out, nothing decoded on the host.
//
// Replaces the reads of the reference's pysam.AlignmentFile (htslib bgzf + zlib inflate + bam_read1) in split_bam's fetch loop
// (workflow/scripts/PreProcessing/SplitBamCellTypes.py:51-124) and behind bam.pileup (SNVCalling/BaseCellCounter.py:190-216):
//   host    walks the BGZF block headers (18 bytes each), copies the file to the device
//   k_inflate      one LANE per BGZF block: raw DEFLATE (inflate_core.h), a wave's 64 decoding tables side by side in LDS; every lane takes
//                  its next block off one queue.  (One WAVE per block with an LDS ring was built and measured: 5-8 x slower, below.)
//                  A block's ISIZE is checked against what it inflates to, its CRC32 by k_block_crc (below): htslib's bgzf reader checks
//                  both, and pysam refuses a file that fails either (SplitBamCellTypes.py:51,65).
//   k_chain        records are a chain (block_size -> next record) through the uncompressed stream; htslib starts every BGZF block on
//                  a record boundary unless a record is longer than a block, so every block's lane walks its own records from the
//                  block's first byte; k_chain_fix hands every block the place where its predecessor's chain really landed, and the
//                  two repeat until nothing moves (one round for an htslib-written file; a file whose records straddle every block
//                  needs a round per block: after LSG_CHAIN_ROUNDS the call gives up and the caller decodes on the host)
//   k_rec_list     every record's offset
//   k_rec_info     one thread per record (bamrec_core.h): validation, CB tag, barcode lookup, SplitBam's counters, shape of its
//                  CIGAR walk (segments, events)
//   k_rec_emit     one WAVE per kept record: lane 0 the read's and segments' words, all lanes the events of every CIGAR operation
//   lsg_load_reads on the device arrays: the tile store (store.hip)
#include "lsg_ctx.h"
#include "inflate_core.h"
#include "cbtable_host.h"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace lsg {

struct IngBlk { uint64_t coff, uoff; uint32_t csize, usize, crc, pad_; };      // crc: the block's CRC32 of its uncompressed bytes (RFC 1952 trailer)

// The decoding tables of a wave's 64 streams (26.9 KB: a byte per symbol, the lit/len symbols' ninth bits apart) live in LDS, five waves
// per CU; the code lengths and the construction's counters, which only the header of a block touches, in global memory (`lens_all`:
// T_LENS * 64 bytes per wave, lane-interleaved like the tables).
__global__ __launch_bounds__(64) void k_inflate(const uint8_t* comp, const IngBlk* blk, uint32_t n_blk, uint8_t* ubuf, uint32_t* status, uint8_t* lens_all) {
    __shared__ uint8_t tab[lsi::T_SYM * 64];
    uint8_t* lens = lens_all + (size_t)blockIdx.x * (lsi::T_LENS * 64);
    const int lane = threadIdx.x;
    // every lane takes its next block from one queue (status[8]) the moment it has finished one: a wave lasts as long as its slowest
    // lane either way, but the blocks of the last, partial round spread over all waves
    for (uint32_t b = blockIdx.x * 64u + (uint32_t)lane; b < n_blk; b = gridDim.x * 64u + atomicAdd(status + 8, 1u)) {
        const IngBlk d = blk[b];
        if (!d.usize) continue;
        const int rc = lsi::inflate_raw(comp + d.coff, d.csize, ubuf + d.uoff, d.usize, lsi::Tab{tab + lane, lens + lane, 64});
        if (rc) { atomicOr(status, 1u); atomicMin(status + 1, b); }
    }
}

// CRC32 (IEEE 802.3, reflected: zlib's crc32) of every block's uncompressed bytes against the block's trailer.  One WAVE per block: lane i
// takes the 1 KB chunk i of the block's <= 64 KB (bytewise table in LDS, 16 bytes per load), the 64 chunk CRCs are combined as zlib's
// crc32_combine does - CRC is linear over GF(2): crc0(A || B) = shift(crc0(A), |B|) ^ crc0(B) with crc0 the register started at 0 and
// shift(v, n) = v run through n zero bytes, done as a product with the precomputed 32 x 32 bit matrices of 2^k zero bytes (zero_ops) -
// and the initial and final complement are put back: crc32(M) = crc0(M) ^ shift(0xffffffff, |M|) ^ 0xffffffff.
struct CrcTables { uint32_t byte_tab[256]; uint32_t zero_ops[17][32]; };      // zero_ops[k][j]: where bit j of the register goes under 2^k zero bytes
__device__ __forceinline__ uint32_t crc_shift(const uint32_t (*ops)[32], uint32_t v, uint32_t n_bytes) {
    for (int k = 0; n_bytes; ++k, n_bytes >>= 1)
        if (n_bytes & 1u) { uint32_t r = 0; for (int j = 0; j < 32; ++j) r ^= (v >> j) & 1u ? ops[k][j] : 0u; v = r; }
    return v;
}
__global__ __launch_bounds__(256) void k_block_crc(const uint8_t* ubuf, const IngBlk* blk, uint32_t n_blk, const CrcTables* tabs, uint32_t* status) {
    __shared__ uint32_t tab[256];
    __shared__ uint32_t ops[17][32];
    for (int i = threadIdx.x; i < 256; i += 256) tab[i] = tabs->byte_tab[i];
    for (int i = threadIdx.x; i < 17 * 32; i += 256) (&ops[0][0])[i] = (&tabs->zero_ops[0][0])[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6); b < n_blk; b += gridDim.x * 4u) {
        const IngBlk d = blk[b];
        if (!d.usize) continue;
        const uint32_t lo = lane * 1024u, hi = lo + 1024u < d.usize ? lo + 1024u : d.usize;
        uint32_t c = 0;
        if (lo < d.usize) {
            const uint8_t* p = ubuf + d.uoff;
            uint32_t i = lo;
            for (; i < hi && ((d.uoff + i) & 15u); ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
            for (; i + 16 <= hi; i += 16) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + i);
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c ^= w[k];
                    c = tab[c & 0xffu] ^ (c >> 8); c = tab[c & 0xffu] ^ (c >> 8); c = tab[c & 0xffu] ^ (c >> 8); c = tab[c & 0xffu] ^ (c >> 8);
                }
            }
            for (; i < hi; ++i) c = tab[(c ^ p[i]) & 0xffu] ^ (c >> 8);
            c = crc_shift(ops, c, d.usize - hi);                      // ... through the bytes of the block behind this chunk
        }
        for (int o = 32; o > 0; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o);
        if (lane == 0) {
            const uint32_t crc = c ^ crc_shift(ops, 0xffffffffu, d.usize) ^ 0xffffffffu;
            if (crc != d.crc) { atomicOr(status, 32u); atomicMin(status + 1, b); }
        }
    }
}
static void make_crc_tables(CrcTables& t) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t.byte_tab[i] = c; }
    for (int j = 0; j < 32; ++j) { uint32_t v = 1u << j; v = t.byte_tab[v & 0xffu] ^ (v >> 8); t.zero_ops[0][j] = v; }      // one zero byte
    for (int k = 1; k < 17; ++k)                                        // the operator of 2^k zero bytes = the one of 2^(k-1) applied twice
        for (int j = 0; j < 32; ++j) { const uint32_t v = t.zero_ops[k - 1][j]; uint32_t r = 0; for (int b = 0; b < 32; ++b) r ^= (v >> b) & 1u ? t.zero_ops[k - 1][b] : 0u; t.zero_ops[k][j] = r; }
}

// (One WAVE per block — uniform decode, a 32 KB LDS ring for the output, match copies spread over the lanes — was built and measured
// this round: 0.8-1.3 s per GB of BAM against 0.17 s for the lane form above.  A DEFLATE symbol is a serial dependency chain of ~100
// scalar-like operations; a wave that runs one chain uses a 64th of the vector unit.  DESIGN.md §8.)

// in[b]: where block b's chain is taken to start (global offset into the uncompressed stream); out: records that start at or after
// in[b] and before the block's end, and where the chain lands at or after that end
// (range: the buffer is a slice of a file — its last record may be cut, which is not an error: lsg_load_bam_range)
__global__ void k_chain(const uint8_t* u, uint64_t total, const IngBlk* blk, uint32_t n_blk, const uint64_t* in, uint64_t* land, uint32_t* nrec, uint32_t* status, int range) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const uint64_t end = blk[b].uoff + blk[b].usize;
    uint64_t p = in[b];
    uint32_t n = 0;
    while (p < end) {
        if (p + 4 > total) { if (!range) atomicOr(status, 2u); break; }
        const uint32_t bs = lsr::rd32(u + p);
        if (bs < 32 || bs > (1u << 30)) { atomicOr(status, 4u); break; }      // (only a wrong guess of in[b] or a corrupt file gets here)
        ++n; p += 4ull + bs;
    }
    land[b] = p; nrec[b] = n;
}
__global__ void k_chain_fix(const IngBlk* blk, uint32_t n_blk, uint64_t first_rec, uint64_t* in, const uint64_t* land, uint32_t* changed) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const uint64_t want = b == 0 ? first_rec : land[b - 1];
    if (in[b] != want) { in[b] = want; atomicAdd(changed, 1u); }
}
__global__ void k_rec_list(const uint8_t* u, const IngBlk* blk, uint32_t n_blk, const uint64_t* in, const uint32_t* rec_base, uint64_t* rec_off) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const uint64_t end = blk[b].uoff + blk[b].usize;
    uint64_t p = in[b];
    uint32_t i = rec_base[b];
    while (p < end) { rec_off[i++] = p; p += 4ull + lsr::rd32(u + p); }
}

struct RecArgs {
    const uint8_t* u; uint64_t total; const uint64_t* rec_off; uint64_t n_rec;
    int32_t n_ref; const int64_t* ref_len; lsr::CbTable cbt; int32_t min_mapq, legacy, keep_unlisted, phased;
    uint8_t* keep; int32_t* cb; uint32_t* nseg; uint32_t* nev;
    unsigned long long* counters;         // total, pass, cb_not_found, cb_not_matched, mapq
    unsigned long long* cb_pass; unsigned long long* cb_low; int64_t n_tally;
    uint32_t* status;
    int32_t range;                        // lsg_load_bam_range: a record cut by the end of the slice is skipped; SplitBam's counters and the tallies take the
    int64_t count_lo, count_hi;           //   records whose (tid << 32 | pos) lies in [count_lo, count_hi) only (the neighbouring slices count the others)
    unsigned long long* last_key;         // largest (tid << 32 | pos) of a complete record (2^63 - 1 for a read without a reference: they end the file)
};
__global__ void k_rec_info(RecArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c_total = 0, c_pass = 0, c_nf = 0, c_nm = 0, c_low = 0, my_key = 0;
    if (i < a.n_rec) {
        const uint64_t p = a.rec_off[i];
        const uint32_t bs = lsr::rd32(a.u + p);
        const uint8_t* rec = a.u + p + 4;
        uint8_t keep = 0; int32_t id = -1; uint32_t nseg = 0, nev = 0;
        if (p + 4ull + bs > a.total) { if (!a.range) atomicOr(a.status, 2u); }
        else {
            const int v = lsr::validate(rec, bs, a.n_ref, a.ref_len);
            const int32_t tid = (int32_t)lsr::rd32(rec);
            const int64_t key = tid >= 0 ? ((int64_t)tid << 32) | (int64_t)(uint32_t)lsr::rd32(rec + 4) : INT64_MAX;
            my_key = (unsigned long long)key;
            const bool mine = key >= a.count_lo && key < a.count_hi;
            if (v != lsr::REC_OK) { atomicOr(a.status, 8u); atomicMin(a.status + 2, (uint32_t)v); }
            else if (tid >= 0) {                                          // infile.fetch() iterates reads placed on a reference
                if (mine) ++c_total;
                const uint32_t mapq = rec[9], n_cigar = lsr::rd16(rec + 12), flag = lsr::rd16(rec + 14);
                uint32_t cb = 0, raw = 0, clean = 0;
                if (!lsr::find_cb(rec, bs, &cb, &raw, &clean)) c_nf += mine;                // read.opt("CB"), SplitBamCellTypes.py:74-79
                else if ((id = lsr::cb_lookup(a.cbt, rec + cb, clean)) < 0) c_nm += mine;   // DICT[barcode], :83-90
                else {
                    const bool low = (int)mapq < a.min_mapq;                                 // report only: the store's load filter / the counts re-apply min_mq
                    if (low) c_low += mine; else c_pass += mine;
                    if (mine && (int64_t)id < a.n_tally) atomicAdd(low ? &a.cb_low[id] : &a.cb_pass[id], 1ull);
                }
                // (a read without a listed barcode stays, with cb = -1, when lsg_set_keep_unlisted asks: the pool of the genotyping pileup)
                if (id >= 0 || a.keep_unlisted) {
                    if (!(flag & 0x4) && n_cigar) {
                        const lsr::Shape sh = lsr::walk<false>(rec, a.legacy, 0, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, a.phased != 0);
                        if (sh.n_events >= (1ull << 31)) atomicOr(a.status, 16u);
                        else { keep = (uint8_t)(1u | (id >= 0 && clean < raw ? 2u : 0u)); nseg = sh.n_segs; nev = (uint32_t)sh.n_events; }
                    }
                }
            }
        }
        a.keep[i] = keep; a.cb[i] = id; a.nseg[i] = nseg; a.nev[i] = nev;
    }
    // SplitBam's counters (SplitBamCellTypes.py:62,117-124): one atomic per wave and counter
    for (int o = 32; o > 0; o >>= 1) {
        c_total += __shfl_down(c_total, o); c_pass += __shfl_down(c_pass, o); c_nf += __shfl_down(c_nf, o); c_nm += __shfl_down(c_nm, o); c_low += __shfl_down(c_low, o);
        const unsigned long long k2 = __shfl_down(my_key, o); my_key = k2 > my_key ? k2 : my_key;
    }
    if ((threadIdx.x & 63) == 0) {
        if (my_key) atomicMax(a.last_key, my_key);
        if (c_total) atomicAdd(&a.counters[0], c_total);
        if (c_pass) atomicAdd(&a.counters[1], c_pass);
        if (c_nf) atomicAdd(&a.counters[2], c_nf);
        if (c_nm) atomicAdd(&a.counters[3], c_nm);
        if (c_low) atomicAdd(&a.counters[4], c_low);
    }
}

struct KeepFlag { const uint8_t* k; __host__ __device__ uint32_t operator()(const uint32_t& i) const { return k[i] & 1u; } };
struct Widen { const uint32_t* v; __host__ __device__ unsigned long long operator()(const uint32_t& i) const { return (unsigned long long)v[i]; } };

struct EmitArgs {
    const uint8_t* u; const uint64_t* rec_off; uint64_t n_rec; int32_t legacy, phased;
    const uint8_t* keep; const int32_t* cb; const uint32_t* ridx; const uint32_t* soff; const unsigned long long* eoff;
    int32_t* read_tid; int32_t* read_pos; uint16_t* read_flag; uint8_t* read_mapq; int32_t* read_cb;
    uint32_t* seg_read; int32_t* seg_start; int32_t* seg_len; int64_t* seg_ev_off; uint16_t* events;
};
__global__ __launch_bounds__(256) void k_rec_emit(EmitArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < a.n_rec; i += n_waves) {
        const uint32_t k = a.keep[i];
        if (!(k & 1u)) continue;
        const uint8_t* rec = a.u + a.rec_off[i] + 4;
        const uint32_t r = a.ridx[i];
        if (lane == 0) {
            const uint32_t flag = lsr::rd16(rec + 14);
            // SAM flags use 12 bits; bit 15 records that the raw CB carried a "-suffix" (the genotyping script looks the RAW tag up)
            a.read_tid[r] = (int32_t)lsr::rd32(rec); a.read_pos[r] = (int32_t)lsr::rd32(rec + 4);
            a.read_flag[r] = (uint16_t)((flag & 0x0fffu) | ((k & 2u) ? LSG_FLAG_CB_SUFFIX : 0u)); a.read_mapq[r] = rec[9]; a.read_cb[r] = a.cb[i];
        }
        const uint32_t s0 = a.soff[i];
        (void)lsr::walk<true>(rec, a.legacy, lane, 64u, r, a.seg_read + s0, a.seg_start + s0, a.seg_len + s0, a.seg_ev_off + s0, (int64_t)a.eoff[i], a.events, a.phased != 0);
    }
}

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }

} // namespace lsg

using namespace lsg;

extern "C" {

static int load_bam_impl(lsg_ctx* c, const uint8_t* file, int64_t n_bytes, int64_t first_record_offset, const char* barcodes, int32_t n_barcodes, const int32_t* ids,
                         int32_t min_mapq, int32_t legacy_del_merge, int range, int64_t count_lo, int64_t count_hi, lsg_bam_info* info, int64_t* cb_pass_out, int64_t* cb_low_out,
                         int64_t n_tally_out) {
    if (!c || !file || n_bytes < 28 || !info || first_record_offset < (range ? 0 : 12)) { set_error("lsg_load_bam: bad arguments"); return -2; }
    if (c->n_contigs <= 0) { set_error("lsg_load_bam: set the contigs (the BAM header's reference table) first"); return -2; }
    if (n_barcodes <= 0) { set_error("lsg_load_bam: a barcode list is needed (the every-CB-is-a-cell mode is the host decoder's)"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const auto t_all = std::chrono::steady_clock::now();
    *info = lsg_bam_info{};
    // ---- the BGZF blocks (every length field checked against the bytes that are there)
    std::vector<IngBlk> blocks;
    uint64_t utotal = 0;
    for (uint64_t off = 0; off < (uint64_t)n_bytes;) {
        if (off + 18 > (uint64_t)n_bytes) { set_error("lsg_load_bam: truncated BGZF block header at offset %llu", (unsigned long long)off); return -1; }
        const uint8_t* h = file + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { set_error("lsg_load_bam: not BGZF (offset %llu)", (unsigned long long)off); return -1; }
        const uint32_t xlen = lsr::rd16(h + 10);
        if (off + 12 + xlen > (uint64_t)n_bytes) { set_error("lsg_load_bam: truncated BGZF extra field at offset %llu", (unsigned long long)off); return -1; }
        uint32_t bsize = 0; bool found = false;
        for (uint32_t q = 0; q + 4 <= xlen;) {
            const uint8_t* sf = h + 12 + q; const uint32_t slen = lsr::rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && q + 6 <= xlen) { bsize = lsr::rd16(sf + 4) + 1u; found = true; }
            q += 4 + slen;
        }
        if (!found || bsize < xlen + 20u || off + bsize > (uint64_t)n_bytes) { set_error("lsg_load_bam: corrupt BGZF block at offset %llu", (unsigned long long)off); return -1; }
        const uint32_t usize = lsr::rd32(h + bsize - 4);
        if (usize > 65536u) { set_error("lsg_load_bam: BGZF block at offset %llu claims %u uncompressed bytes", (unsigned long long)off, usize); return -1; }
        blocks.push_back(IngBlk{off + 12 + xlen, utotal, bsize - xlen - 20, usize, lsr::rd32(h + bsize - 8), 0u});
        utotal += usize; off += bsize;
    }
    const uint32_t n_blk = (uint32_t)blocks.size();
    if ((uint64_t)first_record_offset > utotal) { set_error("lsg_load_bam: the first record lies behind the file's %llu uncompressed bytes", (unsigned long long)utotal); return -1; }
    lsr::CbTableHost cbt; cbt.build(barcodes, n_barcodes, ids);
    const int64_t n_tally = cbt.n_tally;
    // ---- device buffers
    DevBuf d_comp, d_u, d_blk, d_in, d_land, d_nrec, d_base, d_status, d_recoff, d_keep, d_cb, d_nseg, d_nev, d_ridx, d_soff, d_eoff, d_cnt, d_tpass, d_tlow, d_tmp;
    DevBuf d_h, d_id, d_so, d_sl, d_str;
    auto done = [&](int rc) {
        for (DevBuf* b : {&d_comp, &d_u, &d_blk, &d_in, &d_land, &d_nrec, &d_base, &d_status, &d_recoff, &d_keep, &d_cb, &d_nseg, &d_nev, &d_ridx, &d_soff, &d_eoff, &d_cnt, &d_tpass,
                          &d_tlow, &d_tmp, &d_h, &d_id, &d_so, &d_sl, &d_str}) b->release();
        return rc;
    };
    if (d_comp.reserve((size_t)n_bytes + 16) || d_u.reserve((size_t)utotal + 64) || d_blk.reserve((size_t)n_blk * sizeof(IngBlk) + 16) || d_in.reserve(((size_t)n_blk + 1) * 8) ||
        d_land.reserve(((size_t)n_blk + 1) * 8) || d_nrec.reserve(((size_t)n_blk + 2) * 4) || d_base.reserve(((size_t)n_blk + 2) * 4) || d_status.reserve(64) || d_cnt.reserve(64) ||
        d_tpass.reserve(((size_t)n_tally + 1) * 8) || d_tlow.reserve(((size_t)n_tally + 1) * 8) ||
        d_h.reserve(cbt.hash.size() * 8) || d_id.reserve(cbt.id.size() * 4) || d_so.reserve(cbt.str_off.size() * 4) || d_sl.reserve(cbt.str_len.size() * 4) || d_str.reserve(cbt.strs.size() + 16))
        return done(-1);
    hipEvent_t ev[6];
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) { set_error("lsg_load_bam: hipEventCreate failed"); return done(-1); }
    auto done_ev = [&](int rc) { for (auto& e : ev) (void)hipEventDestroy(e); return done(rc); };
#define ING_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return done_ev(-1); } } while (0)
    ING_HIP(hipEventRecord(ev[0], st));
    {   // the file's bytes, through two pinned staging buffers (the caller's memory is usually a read-only file mapping: pageable, and
        // not every runtime can pin it): the CPU fills one while the DMA engine drains the other
        const size_t CH = 32u << 20;
        uint8_t* pin[2] = {nullptr, nullptr}; hipEvent_t pe[2] = {nullptr, nullptr};
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) ok = hipHostMalloc(reinterpret_cast<void**>(&pin[i]), CH, hipHostMallocDefault) == hipSuccess && hipEventCreate(&pe[i]) == hipSuccess;
        size_t off = 0; int slot = 0; bool used[2] = {false, false};
        while (ok && off < (size_t)n_bytes) {
            const size_t n = (size_t)n_bytes - off < CH ? (size_t)n_bytes - off : CH;
            if (used[slot]) ok = hipEventSynchronize(pe[slot]) == hipSuccess;
            if (!ok) break;
            memcpy(pin[slot], file + off, n);
            ok = hipMemcpyAsync(d_comp.as<uint8_t>() + off, pin[slot], n, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(pe[slot], st) == hipSuccess;
            used[slot] = true; off += n; slot ^= 1;
        }
        if (ok) ok = hipStreamSynchronize(st) == hipSuccess;
        for (int i = 0; i < 2; ++i) { if (pe[i]) (void)hipEventDestroy(pe[i]); if (pin[i]) (void)hipHostFree(pin[i]); }
        if (!ok) { set_error("lsg_load_bam: copying the file to the device failed: %s", hipGetErrorString(hipGetLastError())); return done_ev(-1); }
    }
    ING_HIP(hipMemcpyAsync(d_blk.p, blocks.data(), (size_t)n_blk * sizeof(IngBlk), hipMemcpyHostToDevice, st));
    ING_HIP(hipMemcpyAsync(d_h.p, cbt.hash.data(), cbt.hash.size() * 8, hipMemcpyHostToDevice, st));
    ING_HIP(hipMemcpyAsync(d_id.p, cbt.id.data(), cbt.id.size() * 4, hipMemcpyHostToDevice, st));
    ING_HIP(hipMemcpyAsync(d_so.p, cbt.str_off.data(), cbt.str_off.size() * 4, hipMemcpyHostToDevice, st));
    ING_HIP(hipMemcpyAsync(d_sl.p, cbt.str_len.data(), cbt.str_len.size() * 4, hipMemcpyHostToDevice, st));
    ING_HIP(hipMemcpyAsync(d_str.p, cbt.strs.data(), cbt.strs.size(), hipMemcpyHostToDevice, st));
    ING_HIP(hipMemsetAsync(d_status.p, 0, 64, st));
    ING_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_status.as<uint32_t>() + 1), (int)0x7fffffff, 2, st));
    ING_HIP(hipMemsetAsync(d_cnt.p, 0, 64, st));
    ING_HIP(hipMemsetAsync(d_tpass.p, 0, ((size_t)n_tally + 1) * 8, st));
    ING_HIP(hipMemsetAsync(d_tlow.p, 0, ((size_t)n_tally + 1) * 8, st));
    ING_HIP(hipEventRecord(ev[1], st));
    // ---- inflate
    const IngBlk* dblk = d_blk.as<IngBlk>();
    uint32_t* status = d_status.as<uint32_t>();
    {                                                        // a lane per block; 26.9 KB of LDS tables per wave: five waves per CU
        static const unsigned per_cu = getenv("LSG_INFLATE_WAVES") ? (unsigned)atoi(getenv("LSG_INFLATE_WAVES")) : 5u;
        unsigned g = (n_blk + 63) / 64; const unsigned cap = (unsigned)c->n_cus * (per_cu ? per_cu : 5u);
        if (g > cap) g = cap;
        if (!g) g = 1;
        if (d_tmp.reserve((size_t)g * lsi::T_LENS * 64)) return done_ev(-3);
        hipLaunchKernelGGL(k_inflate, dim3(g), dim3(64), 0, st, d_comp.as<uint8_t>(), dblk, n_blk, d_u.as<uint8_t>(), status, d_tmp.as<uint8_t>());
    }
    ING_HIP(hipEventRecord(ev[2], st));
    uint32_t hstat[4] = {0, 0, 0, 0};
    ING_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, st));
    ING_HIP(hipStreamSynchronize(st));
    if (hstat[0] & 1u) { set_error("lsg_load_bam: inflate failed in BGZF block %u", hstat[1]); return done_ev(-1); }
    d_comp.release();
    if (!getenv("LSG_NO_BGZF_CRC")) {                     // every block's CRC32 against its trailer (k_block_crc)
        static CrcTables h_tabs; static bool h_tabs_made = false;
        if (!h_tabs_made) { make_crc_tables(h_tabs); h_tabs_made = true; }
        DevBuf d_tabs;
        if (d_tabs.reserve(sizeof(CrcTables))) return done_ev(-1);
        ING_HIP(hipMemcpyAsync(d_tabs.p, &h_tabs, sizeof(CrcTables), hipMemcpyHostToDevice, st));
        ING_HIP(hipMemsetAsync(status, 0, 4, st));
        ING_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(status + 1), (int)0x7fffffff, 1, st));
        unsigned g = (n_blk + 3) / 4; const unsigned cap = (unsigned)(c->n_cus * 8);
        if (g > cap) g = cap;
        hipLaunchKernelGGL(k_block_crc, dim3(g ? g : 1), dim3(256), 0, st, d_u.as<uint8_t>(), dblk, n_blk, d_tabs.as<CrcTables>(), status);
        ING_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, st));
        const hipError_t es = hipStreamSynchronize(st);
        d_tabs.release();
        if (es != hipSuccess) { set_error("lsg_load_bam: the CRC pass failed: %s", hipGetErrorString(es)); return done_ev(-1); }
        if (hstat[0] & 32u) { set_error("lsg_load_bam: CRC32 mismatch in BGZF block %u", hstat[1]); return done_ev(-1); }
    }
    // ---- the record chain
    const uint8_t* u = d_u.as<uint8_t>();
    uint64_t* in = d_in.as<uint64_t>(); uint64_t* land = d_land.as<uint64_t>();
    {
        std::vector<uint64_t> h_in(n_blk);
        for (uint32_t b = 0; b < n_blk; ++b) h_in[b] = b == 0 ? (uint64_t)first_record_offset : blocks[b].uoff;      // htslib: a block starts on a record boundary
        ING_HIP(hipMemcpyAsync(in, h_in.data(), (size_t)n_blk * 8, hipMemcpyHostToDevice, st));
        ING_HIP(hipStreamSynchronize(st));
    }
    const int max_rounds = env_int("LSG_CHAIN_ROUNDS", 512);
    int rounds = 0;
    uint32_t* changed = status + 3;
    for (;; ++rounds) {
        if (rounds > max_rounds) {
            set_error("lsg_load_bam: the records of this BAM straddle its BGZF blocks (not written by htslib's bam_write1?): %d rounds did not settle the record chain; decode it on the host", max_rounds);
            return done_ev(-4);
        }
        ING_HIP(hipMemsetAsync(status, 0, 4, st));
        ING_HIP(hipMemsetAsync(changed, 0, 4, st));
        hipLaunchKernelGGL(k_chain, dim3((n_blk + 255) / 256), dim3(256), 0, st, u, utotal, dblk, n_blk, in, land, d_nrec.as<uint32_t>(), status, range);
        hipLaunchKernelGGL(k_chain_fix, dim3((n_blk + 255) / 256), dim3(256), 0, st, dblk, n_blk, (uint64_t)first_record_offset, in, land, changed);
        ING_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
        if (hstat[3] == 0) break;                      // every block started where its predecessor's chain landed: the walk just made is the true one
    }
    if (hstat[0] & 6u) { set_error("lsg_load_bam: %s", (hstat[0] & 2u) ? "truncated record at the end of the file" : "record with an impossible block_size"); return done_ev(-1); }
    {
        uint64_t last_land = 0;
        ING_HIP(hipMemcpyAsync(&last_land, land + (n_blk - 1), 8, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
        if (!range && last_land != utotal) { set_error("lsg_load_bam: truncated record at the end of the file"); return done_ev(-1); }
    }
    ING_HIP(hipEventRecord(ev[3], st));
    uint32_t n_rec32 = 0;
    {
        ING_HIP(hipMemsetAsync(d_nrec.as<uint32_t>() + n_blk, 0, 4, st));
        size_t tb = 0;
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_nrec.as<uint32_t>(), d_base.as<uint32_t>(), (int)(n_blk + 1), st));
        if (d_tmp.reserve(tb + 256)) return done_ev(-1);
        tb = d_tmp.cap;
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, d_nrec.as<uint32_t>(), d_base.as<uint32_t>(), (int)(n_blk + 1), st));
        ING_HIP(hipMemcpyAsync(&n_rec32, d_base.as<uint32_t>() + n_blk, 4, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
    }
    // the events in the tile-phased layout (LSG_LAYOUT_PHASED): the count then fetches every entry as ONE 128-byte line; LSG_INGEST_COMPACT=1: segment after segment
    const int32_t phased = getenv("LSG_INGEST_COMPACT") ? 0 : 1;
    const uint64_t n_rec = n_rec32;
    if (n_rec >= 0x7fffff00ull) { set_error("lsg_load_bam: more than 2^31 records; load the file in windows on the host"); return done_ev(-2); }
    if (d_recoff.reserve((n_rec + 1) * 8) || d_keep.reserve(n_rec + 16) || d_cb.reserve((n_rec + 1) * 4) || d_nseg.reserve((n_rec + 2) * 4) || d_nev.reserve((n_rec + 2) * 4) ||
        d_ridx.reserve((n_rec + 2) * 4) || d_soff.reserve((n_rec + 2) * 4) || d_eoff.reserve((n_rec + 2) * 8)) return done_ev(-1);
    uint32_t R = 0, S = 0; unsigned long long E = 0;
    if (n_rec) {
        hipLaunchKernelGGL(k_rec_list, dim3((n_blk + 255) / 256), dim3(256), 0, st, u, dblk, n_blk, in, d_base.as<uint32_t>(), d_recoff.as<uint64_t>());
        RecArgs ra{};
        ra.u = u; ra.total = utotal; ra.rec_off = d_recoff.as<uint64_t>(); ra.n_rec = n_rec; ra.n_ref = c->n_contigs; ra.ref_len = c->d_contig_len.as<int64_t>();
        ra.cbt = lsr::CbTable{d_h.as<uint64_t>(), d_id.as<int32_t>(), d_so.as<uint32_t>(), d_sl.as<uint32_t>(), d_str.as<uint8_t>(), cbt.mask};
        ra.min_mapq = min_mapq; ra.legacy = legacy_del_merge ? 1 : 0; ra.keep_unlisted = c->keep_unlisted ? 1 : 0; ra.phased = phased;
        ra.keep = d_keep.as<uint8_t>(); ra.cb = d_cb.as<int32_t>(); ra.nseg = d_nseg.as<uint32_t>(); ra.nev = d_nev.as<uint32_t>();
        ra.counters = d_cnt.as<unsigned long long>(); ra.cb_pass = d_tpass.as<unsigned long long>(); ra.cb_low = d_tlow.as<unsigned long long>(); ra.n_tally = n_tally;
        ra.status = status;
        ra.range = range; ra.count_lo = count_lo; ra.count_hi = count_hi; ra.last_key = d_cnt.as<unsigned long long>() + 6;
        ING_HIP(hipMemsetAsync(status, 0, 4, st));
        hipLaunchKernelGGL(k_rec_info, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, ra);
        ING_HIP(hipMemcpyAsync(hstat, status, 16, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
        if (hstat[0]) {
            static const char* why[] = {"", "a record shorter than its fixed fields", "a record whose fields exceed its block_size", "a record on a reference the header does not list",
                                        "a CIGAR operation above 8", "a CIGAR that does not cover the stored sequence", "an alignment that leaves its reference"};
            set_error("lsg_load_bam: %s", (hstat[0] & 8u) && hstat[2] < 7 ? why[hstat[2]] : (hstat[0] & 16u) ? "a read of 2^31 or more pileup events" : "truncated record at the end of the file");
            return done_ev(-1);
        }
        // places of the kept records in the read / segment / event arrays
        hipcub::CountingInputIterator<uint32_t> iota(0);
        {
            hipcub::TransformInputIterator<uint32_t, KeepFlag, hipcub::CountingInputIterator<uint32_t>> it(iota, KeepFlag{d_keep.as<uint8_t>()});
            size_t tb = 0;
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, d_ridx.as<uint32_t>(), (int)(n_rec + 1), st));
            if (d_tmp.reserve(tb + 256)) return done_ev(-1);
            tb = d_tmp.cap;
            ING_HIP(hipMemsetAsync(d_keep.as<uint8_t>() + n_rec, 0, 1, st));
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, it, d_ridx.as<uint32_t>(), (int)(n_rec + 1), st));
        }
        {
            ING_HIP(hipMemsetAsync(d_nseg.as<uint32_t>() + n_rec, 0, 4, st));
            ING_HIP(hipMemsetAsync(d_nev.as<uint32_t>() + n_rec, 0, 4, st));
            size_t tb = 0;
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_nseg.as<uint32_t>(), d_soff.as<uint32_t>(), (int)(n_rec + 1), st));
            if (d_tmp.reserve(tb + 256)) return done_ev(-1);
            tb = d_tmp.cap;
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, d_nseg.as<uint32_t>(), d_soff.as<uint32_t>(), (int)(n_rec + 1), st));
            hipcub::TransformInputIterator<unsigned long long, Widen, hipcub::CountingInputIterator<uint32_t>> it(iota, Widen{d_nev.as<uint32_t>()});
            tb = 0;
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, d_eoff.as<unsigned long long>(), (int)(n_rec + 1), st));
            if (d_tmp.reserve(tb + 256)) return done_ev(-1);
            tb = d_tmp.cap;
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, it, d_eoff.as<unsigned long long>(), (int)(n_rec + 1), st));
        }
        ING_HIP(hipMemcpyAsync(&R, d_ridx.as<uint32_t>() + n_rec, 4, hipMemcpyDeviceToHost, st));
        ING_HIP(hipMemcpyAsync(&S, d_soff.as<uint32_t>() + n_rec, 4, hipMemcpyDeviceToHost, st));
        ING_HIP(hipMemcpyAsync(&E, d_eoff.as<unsigned long long>() + n_rec, 8, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
    }
    // ---- the read-record arrays, on the device
    DevBuf &o_tid = c->gen[0], &o_pos = c->gen[1], &o_flag = c->gen[2], &o_mapq = c->gen[3], &o_cb = c->gen[4], &o_sread = c->gen[5], &o_sstart = c->gen[6],
           &o_slen = c->gen[7], &o_sevoff = c->gen[8], &o_events = c->gen[9];
    if (o_tid.reserve(((size_t)R + 1) * 4) || o_pos.reserve(((size_t)R + 1) * 4) || o_flag.reserve(((size_t)R + 1) * 2) || o_mapq.reserve((size_t)R + 1) || o_cb.reserve(((size_t)R + 1) * 4) ||
        o_sread.reserve(((size_t)S + 1) * 4) || o_sstart.reserve(((size_t)S + 1) * 4) || o_slen.reserve(((size_t)S + 1) * 4) || o_sevoff.reserve(((size_t)S + 1) * 8) ||
        o_events.reserve(((size_t)E + 1) * 2)) return done_ev(-1);
    if (phased && E) ING_HIP(hipMemsetAsync(o_events.p, 0, (size_t)E * 2, st));      // (the gaps between the segments hold 0)
    if (R) {
        EmitArgs ea{};
        ea.u = u; ea.rec_off = d_recoff.as<uint64_t>(); ea.n_rec = n_rec; ea.legacy = legacy_del_merge ? 1 : 0; ea.phased = phased;
        ea.keep = d_keep.as<uint8_t>(); ea.cb = d_cb.as<int32_t>(); ea.ridx = d_ridx.as<uint32_t>(); ea.soff = d_soff.as<uint32_t>(); ea.eoff = d_eoff.as<unsigned long long>();
        ea.read_tid = o_tid.as<int32_t>(); ea.read_pos = o_pos.as<int32_t>(); ea.read_flag = o_flag.as<uint16_t>(); ea.read_mapq = o_mapq.as<uint8_t>(); ea.read_cb = o_cb.as<int32_t>();
        ea.seg_read = o_sread.as<uint32_t>(); ea.seg_start = o_sstart.as<int32_t>(); ea.seg_len = o_slen.as<int32_t>(); ea.seg_ev_off = o_sevoff.as<int64_t>(); ea.events = o_events.as<uint16_t>();
        unsigned g = (unsigned)((n_rec + 3) / 4); const unsigned cap = (unsigned)(c->n_cus * 32);
        if (g > cap) g = cap;
        hipLaunchKernelGGL(k_rec_emit, dim3(g), dim3(256), 0, st, ea);
    }
    ING_HIP(hipEventRecord(ev[4], st));
    unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ING_HIP(hipMemcpyAsync(cnt, d_cnt.p, 56, hipMemcpyDeviceToHost, st));
    if (cb_pass_out && cb_low_out) {
        const int64_t n_copy = n_tally < n_tally_out ? n_tally : n_tally_out;
        for (int64_t i = 0; i < n_tally_out; ++i) { cb_pass_out[i] = 0; cb_low_out[i] = 0; }
        if (n_copy > 0) {
            ING_HIP(hipMemcpyAsync(cb_pass_out, d_tpass.p, (size_t)n_copy * 8, hipMemcpyDeviceToHost, st));
            ING_HIP(hipMemcpyAsync(cb_low_out, d_tlow.p, (size_t)n_copy * 8, hipMemcpyDeviceToHost, st));
        }
    }
    ING_HIP(hipGetLastError());
    ING_HIP(hipStreamSynchronize(st));
    float ms[4] = {0, 0, 0, 0};
    (void)hipEventElapsedTime(&ms[0], ev[0], ev[1]); (void)hipEventElapsedTime(&ms[1], ev[1], ev[2]); (void)hipEventElapsedTime(&ms[2], ev[2], ev[3]); (void)hipEventElapsedTime(&ms[3], ev[3], ev[4]);
    // the uncompressed stream and the per-record words are done with before the store is built
    for (DevBuf* b : {&d_u, &d_recoff, &d_keep, &d_cb, &d_nseg, &d_nev, &d_ridx, &d_soff, &d_eoff}) b->release();
    lsg_reads g{};
    g.n_reads = R; g.n_segs = S; g.n_events = (int64_t)E; g.on_device = 1;
    g.read_tid = o_tid.as<int32_t>(); g.read_pos = o_pos.as<int32_t>(); g.read_flag = o_flag.as<uint16_t>(); g.read_mapq = o_mapq.as<uint8_t>(); g.read_cb = o_cb.as<int32_t>();
    g.seg_read = o_sread.as<uint32_t>(); g.seg_start = o_sstart.as<int32_t>(); g.seg_len = o_slen.as<int32_t>(); g.seg_ev_off = o_sevoff.as<int64_t>(); g.events = o_events.as<uint16_t>();
    c->hint_phased_events = phased ? o_events.p : nullptr;
    const auto t_store = std::chrono::steady_clock::now();
    const int rc = lsg_load_reads(c, &g);
    c->hint_phased_events = nullptr;
    for (auto& b : c->gen) b.release();
    if (rc) return done_ev(rc);
    info->total_reads = (int64_t)cnt[0]; info->pass_reads = (int64_t)cnt[1]; info->cb_not_found = (int64_t)cnt[2]; info->cb_not_matched = (int64_t)cnt[3]; info->mapq_filtered = (int64_t)cnt[4];
    info->n_blocks = n_blk; info->n_records = (int64_t)n_rec; info->n_ubytes = (int64_t)utotal; info->chain_rounds = rounds + 1;
    info->ms_h2d = ms[0]; info->ms_inflate = ms[1]; info->ms_chain = ms[2]; info->ms_decode = ms[3];
    info->ms_store = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_store).count();
    info->ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_all).count();
    info->last_key = n_rec ? (int64_t)cnt[6] : -1;
    return done_ev(0);
#undef ING_HIP
}

int lsg_load_bam(lsg_ctx* c, const uint8_t* file, int64_t n_bytes, int64_t first_record_offset, const char* barcodes, int32_t n_barcodes, const int32_t* ids,
                 int32_t min_mapq, int32_t legacy_del_merge, lsg_bam_info* info, int64_t* cb_pass_out, int64_t* cb_low_out, int64_t n_tally_out) {
    return load_bam_impl(c, file, n_bytes, first_record_offset, barcodes, n_barcodes, ids, min_mapq, legacy_del_merge, 0, INT64_MIN, INT64_MAX, info, cb_pass_out, cb_low_out, n_tally_out);
}

int lsg_load_bam_range(lsg_ctx* c, const uint8_t* slice, int64_t n_bytes, int64_t first_record_offset, const char* barcodes, int32_t n_barcodes, const int32_t* ids,
                       int32_t min_mapq, int32_t legacy_del_merge, int64_t count_lo_key, int64_t count_hi_key, lsg_bam_info* info, int64_t* cb_pass_out, int64_t* cb_low_out,
                       int64_t n_tally_out) {
    return load_bam_impl(c, slice, n_bytes, first_record_offset, barcodes, n_barcodes, ids, min_mapq, legacy_del_merge, 1, count_lo_key, count_hi_key, info, cb_pass_out, cb_low_out, n_tally_out);
}

} // extern "C"
