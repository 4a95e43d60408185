// The tile store: what lsg_load_reads leaves in HBM (lsg_ctx.h).
//
// A pileup IS the transposition of read-major events into column order.  The reference does it per 50 kb window with htslib's
// bam_plp behind bam.pileup(...) (workflow/scripts/SNVCalling/BaseCellCounter.py:190-198); here it is done ONCE per load, on the
// device, straight from the caller's compact read-record arrays:
//   1. capacities   every (segment x 64-position tile) overlap of a read that carries a barcode is one ENTRY; a segment's entries are a
//                   contiguous range of tiles, so k_seg_static marks +1 at its first tile and -1 past its last, a running sum gives
//                   every tile's number of entries and a second scan its region
//   2. scatter      k_bin writes a 16-byte record per entry into its tile's region, in arrival order, beside its sort key (barcode)
//   3. sort         the entries of each tile by barcode (segmented radix sort over the tile regions): equal barcodes become adjacent
//                   RUNS, which is what turns len(set(cells)) (BaseCellCounter.py:283,292) into "entries minus duplicates in a run"
//   4. fill         per entry, in that order: barcode / strand / run flags, events - 1, SAM flag and MAPQ (admission is decided per
//                   count), owning read, and where its events lie in the caller's array
//   5. gather       eight entries to a 1 KB block held transposed ([position 0..63][entry 0..7]): each entry's <= 128 bytes arrive by
//                   16-byte loads (eight lanes per entry: one wave-instruction fetches a whole block), cross an LDS tile and leave as
//                   one kilobyte store per block; the block's extent (first / last position with an event) beside it
// Nothing here depends on count parameters or on the barcode -> cell-type table, so the caller's events are not needed again: the
// store is the only resident copy (lsg_set_keep_reads keeps them for tests).  The plan of a count over the store (jobs, units, slabs:
// ensure_plan) depends on the number of cell types only.
#include "lsg_ctx.h"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <utility>

namespace lsg {

constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

// An entry on its way through the sort.  Key (64 bits): barcode [0 .. cb_bits) | first position in the tile, 6 bits | events - 1, 6 bits |
// offset of its first event in the caller's array, the remaining 52 - cb_bits bits (build_store checks that the events fit).  Only the
// barcode bits are sorted on.  Value (32 bits): owning read [0 .. 29) | upper pileup window of its tile << 29 | first entry of its segment << 30 |
// forward strand << 31.
constexpr uint32_t RV_WHI = 1u << 29, RV_SEGFIRST = 1u << 30, RV_FWD = 1u << 31, RV_READ = RV_WHI - 1u;
// (wsh = 1: the bins are 128-position windows - first position and events - 1 take seven bits each)
__host__ __device__ __forceinline__ uint64_t sort_key(uint32_t cb, uint32_t first, uint32_t nev1, uint64_t src, int cb_bits, int wsh = 0) {
    return (uint64_t)cb | ((uint64_t)(first | (nev1 << (6 + wsh))) << cb_bits) | (src << (cb_bits + 12 + 2 * wsh));
}

struct BuildArgs {
    int64_t n_reads, n_segs, n_events;
    const int32_t* read_tid; const uint16_t* read_flag; const uint8_t* read_mapq; const int32_t* read_cb;
    const uint32_t* seg_read; const int32_t* seg_start; const int32_t* seg_len; const int64_t* seg_ev_off;
    const uint32_t* tile_base; const int64_t* contig_len;
    int32_t n_contigs; uint32_t n_tiles;
    uint2* seg_info;                      // per segment {barcode | reverse << 24, or KEY_INVALID; first tile of its contig}
    uint32_t* tile_cap;                   // entries per tile
    uint32_t* cursor;                     // next free place of every tile's region
    uint64_t* key; uint32_t* rdv;         // per entry: the packed sort key (barcode in its low cb_bits bits, below) and the owning read | flags
    int32_t cb_bits;                      // bits of the largest barcode id
    unsigned long long* qhead;            // work queue head of the binning pass
    int32_t lf_min_mq, lf_ignore_orphans; uint32_t lf_flag_exclude;      // the load filter (lsg_set_load_filter)
    uint32_t* bad;                        // bit 0: a segment's event range lies outside the events; bit 1: a segment's read index outside the reads; bit 2: an admitted segment whose events are not tile-phased ((seg_ev_off - seg_start) % 64 != 0)
    unsigned long long* n_ev;             // events of the statically admitted segments = events the store will hold
    int32_t* span_diff;                   // [n_tiles + 1] marks of the reads' spans (the depth cap's bound), or null
    int32_t* cap_diff;                    // [n_tiles + 1] marks of the admitted segments' tile ranges: +1 at the first tile, -1 past the last; their running sum = entries per tile
    int32_t window;                       // the reference's pileup windows [1 + k W, 1 + (k + 1) W): an entry never crosses an edge of one
    int32_t wsh;                          // 0: the bins the entries are sorted into are the 64-position tiles; 1: 128-position WINDOWS, tiles (2 w, 2 w + 1) - a load that keeps
                                          // no store over events phased modulo 128: an entry is then one aligned 256-byte block, which the memory system delivers at
                                          // 23 G blocks/s where it delivers single 128-byte lines at 29 G/s (tools/block_rate.hip), and there are 0.58 x as many
    int32_t src_shift;                    // 6 when the caller's events are tile-phased (LSG_LAYOUT_PHASED: the key's source field is the entry's 128-byte LINE, its low six bits being the entry's first position), else 0
};

// The edges of the reference's pileup windows (BaseCellCounter.py:81-113: [1, 50001), [50001, 100001), ...) that lie INSIDE a tile cut
// the segments crossing them: the tile gets two entries of such a segment, one per window, because with max_depth every window is a
// pileup of its own (:185-191) and a read may be dropped in one and counted in the next.  First edge after position x:
// (positions are below 2^31: one 32-bit division)
__device__ __forceinline__ int64_t win_edge_after(int64_t x, int32_t W) { return 1 + (int64_t)W * (int64_t)((x >= 1 ? (uint32_t)(x - 1) / (uint32_t)W : 0u) + 1u); }
// the edge strictly inside the tile that starts at tstart (W >= 64: at most one), or -1
__device__ __forceinline__ int64_t win_edge_in_tile(int64_t tstart, int32_t W) {
    const int64_t b = win_edge_after(tstart, W);
    return b < tstart + TILE_W ? b : -1;
}

// Static admission of a segment: its read carries a barcode, passes the load filter and lies on a contig, the segment lies inside the
// contig (what ANY count parameters or barcode table can admit; malformed segments are never counted).  Also the load's validation of
// the caller's arrays — and the marks of the depth cap's bound (layout.hip): +1 at the tile a read's first segment starts in, -1 past
// the tile its last segment ends in, for every read with a barcode.  The reads of a deep gene start and end in the same few tiles and a
// word takes ~90 atomics per microsecond, so a workgroup merges the marks of a batch of 256 consecutive segments in an LDS hash and
// issues one global atomic per distinct tile.
// The tiles' CAPACITIES come out of the same pass the same way: a segment's entries are the tiles of one contiguous range, so +1 at its
// first tile and -1 past its last one, summed along the tiles, is the number of entries of every tile — two marks per segment where a
// counting pass over the entries (0.9 ms of LDS atomics for C2's 185 M entries) made one per entry.
// The pass waits for memory (a segment's arrays, then its read's, then its contig's: three dependent trips; 84 % of its wave-cycles were
// spent waiting with one segment per thread): a thread takes SEG_U segments of a batch and issues each level's loads for all of them
// before it looks at any.
constexpr int SEG_THREADS = 256, SEG_U = 4, SEG_H = 2048, SEG_MAX_SINCE = 6;       // (6 x 2048 marks of one sign at most per half word)
__global__ __launch_bounds__(SEG_THREADS) void k_seg_static(BuildArgs a) {
    __shared__ uint32_t hkey[SEG_H];
    __shared__ int32_t hval[SEG_H];       // both sums of a tile in one word: span marks in the low half, capacity marks x 65536 (each at most 2048 in size per batch, SEG_MAX_SINCE batches per flush)
    __shared__ unsigned long long s_ev;
    __shared__ uint32_t s_new, s_flush;        // tiles in the hash since its last flush; this batch ends with one
    for (int i = threadIdx.x; i < SEG_H; i += SEG_THREADS) { hkey[i] = KEY_INVALID; hval[i] = 0; }
    if (threadIdx.x == 0) { s_ev = 0; s_new = 0; s_flush = 0; }
    // a tile's slot in the hash, or -1 when sixteen probes find none (a batch marks 4096 tiles at worst, the hash holds 2048 and is
    // flushed when half full: the marks of a batch that scatters that widely go straight to memory, as they would from a full hash)
    auto slot = [&](uint32_t t) -> int {
        uint32_t h = (t * 2654435761u) >> 21;
        for (int tries = 0; tries < 16; ++tries) {
            const uint32_t prev = atomicCAS(&hkey[h], KEY_INVALID, t);
            if (prev == KEY_INVALID) { atomicAdd(&s_new, 1u); return (int)h; }
            if (prev == t) return (int)h;
            h = (h + 1) & (SEG_H - 1);
        }
        return -1;
    };
    auto mark = [&](uint32_t t, int32_t v) { const int h = slot(t); if (h >= 0) atomicAdd(&hval[h], v); else atomicAdd(a.span_diff + t, v); };
    auto mark_cap = [&](uint32_t t, int32_t v) { const int h = slot(t); if (h >= 0) atomicAdd(&hval[h], v * 65536); else atomicAdd(a.cap_diff + t, v); };
    unsigned long long n_ev = 0;
    bool unphased = false;
    // A workgroup takes CONSECUTIVE batches (the segments of a coordinate-sorted BAM arrive gene by gene: the next batch marks the same
    // few tiles) and flushes the hash only when the next batch might not fit any more, or after SEG_MAX_SINCE batches (the packed sums
    // stay inside their 16 bits), or at its end.
    constexpr int64_t BATCH = (int64_t)SEG_THREADS * SEG_U;
    const int64_t n_batches = (a.n_segs + BATCH - 1) / BATCH;
    const int64_t per_wg = (n_batches + gridDim.x - 1) / gridDim.x;
    const int64_t b_lo = (int64_t)blockIdx.x * per_wg, b_hi = b_lo + per_wg < n_batches ? b_lo + per_wg : n_batches;
    const int64_t last_seg = a.n_segs - 1;
    int since = 0;
    __syncthreads();
    for (int64_t bt = b_lo; bt < b_hi; ++bt) {
        // level 1: the segments' own arrays (a segment past the end reads the last one and is not looked at)
        int64_t sg[SEG_U], st_[SEG_U], ln_[SEG_U], of_[SEG_U]; uint32_t rd_[SEG_U], rp_[SEG_U], rn_[SEG_U]; bool in_[SEG_U];
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {
            sg[u] = bt * BATCH + (int64_t)u * SEG_THREADS + threadIdx.x;
            in_[u] = sg[u] < a.n_segs;
            const int64_t si = in_[u] ? sg[u] : last_seg;
            rd_[u] = a.seg_read[si]; st_[u] = a.seg_start[si]; ln_[u] = a.seg_len[si]; of_[u] = a.seg_ev_off[si];
            rp_[u] = a.seg_read[si > 0 ? si - 1 : 0]; rn_[u] = a.seg_read[si < last_seg ? si + 1 : last_seg];
        }
        // level 2: their reads'
        int32_t tid_[SEG_U], cb_[SEG_U]; uint32_t flag_[SEG_U], mq_[SEG_U];
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {
            const int64_t ri = (int64_t)rd_[u] < a.n_reads ? (int64_t)rd_[u] : 0;
            tid_[u] = a.read_tid[ri]; cb_[u] = a.read_cb[ri]; flag_[u] = a.read_flag[ri]; mq_[u] = (uint32_t)a.read_mapq[ri];
        }
        // level 3: their contigs'
        int64_t clen_[SEG_U]; uint32_t tb_[SEG_U], te_[SEG_U];
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {
            const int ti = tid_[u] >= 0 && tid_[u] < a.n_contigs ? tid_[u] : 0;
            clen_[u] = a.contig_len[ti]; tb_[u] = a.tile_base[ti]; te_[u] = a.tile_base[ti + 1];
        }
#pragma unroll
        for (int u = 0; u < SEG_U; ++u) {
        const int64_t s = sg[u];
        uint32_t edge_tile = KEY_INVALID;                   // tile of the first window edge inside this thread's segment
        if (in_[u]) {
            const uint32_t r = rd_[u];
            uint32_t key = KEY_INVALID, tb = 0;
            if ((int64_t)r >= a.n_reads) atomicOr(a.bad, 2u);
            else {
                const int64_t st = st_[u], ln = ln_[u], o = of_[u];
                const int32_t tid = tid_[u], cb = cb_[u];
                const bool on_contig = tid >= 0 && tid < a.n_contigs;
                if (ln > 0 && (o < 0 || o + ln > a.n_events)) atomicOr(a.bad, 1u);
                else {
                    const uint32_t flag = flag_[u];
                    // the load filter: what SplitBamCellTypes.py:110-113 does to the BAM before BaseCellCounter ever sees it
                    bool pool = (int)mq_[u] >= a.lf_min_mq && (flag & a.lf_flag_exclude) == 0;
                    if (pool && a.lf_ignore_orphans && (flag & 1u) && !(flag & 2u)) pool = false;
                    if (pool && on_contig && cb >= 0 && (uint32_t)cb < CB_MASK && !(st < 0 || ln <= 0 || st + ln > clen_[u])) {
                        key = (uint32_t)cb | (((flag >> 4) & 1u) << 24);
                        tb = tb_[u];
                        n_ev += (unsigned long long)ln;
                        unphased |= ((o - st) & (int64_t)((64 << a.wsh) - 1)) != 0;
                        mark_cap((tb + ((uint32_t)st >> 6)) >> a.wsh, 1);
                        mark_cap(((tb + ((uint32_t)(st + ln - 1) >> 6)) >> a.wsh) + 1, -1);
                        // one more entry in the tile of every window edge strictly inside the segment: rare (one segment in forty), so not
                        // through the hash - the first edge's tile is handed to the wave below (neighbouring segments of a deep gene cross
                        // the SAME edge: one atomic per distinct tile and wave), further ones (a segment longer than a window) straight to memory
                        int nb = 0;
                        for (int64_t b = win_edge_after(st, a.window); b < st + ln; b += a.window) {
                            if ((b & (int64_t)((64 << a.wsh) - 1)) == 0) continue;       // (an edge on a bin's boundary cuts nothing)
                            const uint32_t t = (tb + (uint32_t)(b >> 6)) >> a.wsh;
                            if (nb++ == 0) edge_tile = t;
                            else { atomicAdd(a.cap_diff + t, 1); atomicAdd(a.cap_diff + t + 1, -1); }
                        }
                    }
                }
                if (a.span_diff && on_contig && cb >= 0) {          // (every read with a barcode, whatever the load filter: a bound never under-counts)
                    const bool first = s == 0 || rp_[u] != r, last = s + 1 == a.n_segs || rn_[u] != r;
                    const uint32_t t0 = tb_[u], te = te_[u];
                    if ((first || last) && te > t0) {
                        // the read is still buffered while the column AFTER its last one is entered (freed by that column's sweep): span end inclusive;
                        // both marks are clamped into the contig so that every +1 has its -1
                        int64_t b = st < 0 ? 0 : st, e = st + (ln > 0 ? ln : 0);
                        if (e < 0) e = 0;
                        if (first) { uint32_t t = t0 + (uint32_t)(b >> 6); if (t >= te) t = te - 1; mark(t >> a.wsh, 1); }
                        if (last) { uint32_t t = t0 + (uint32_t)(e >> 6); if (t >= te) t = te - 1; mark((t >> a.wsh) + 1, -1); }      // (past the last bin it touches)
                    }
                }
            }
            a.seg_info[s] = make_uint2(key, tb);
        }
        for (unsigned long long todo = __ballot(edge_tile != KEY_INVALID); todo;) {      // (every lane of the wave is here)
            const uint32_t lt = (uint32_t)__shfl((int)edge_tile, __ffsll((long long)todo) - 1);
            const unsigned long long same = __ballot(edge_tile == lt);
            if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)todo) - 1)) { const int n = __popcll(same); atomicAdd(a.cap_diff + lt, n); atomicAdd(a.cap_diff + lt + 1, -n); }
            todo &= ~same;
        }
        }
        ++since;
        __syncthreads();                                  // the batch's marks are in
        if (threadIdx.x == 0) {
            s_flush = (bt + 1 == b_hi || s_new > SEG_H / 2 || since >= SEG_MAX_SINCE) ? 1u : 0u;
            if (s_flush) s_new = 0;
        }
        __syncthreads();                                  // every thread sees the same decision; nobody marks meanwhile
        if (s_flush) {
            since = 0;
            for (int i = threadIdx.x; i < SEG_H; i += SEG_THREADS)
                if (hkey[i] != KEY_INVALID) {
                    const int32_t w = hval[i], v = (int32_t)(int16_t)(w & 0xffff), cp = (w - v) >> 16;       // (w = cp * 65536 + v exactly)
                    if (v != 0) atomicAdd(a.span_diff + hkey[i], v);       // (marks of the spans are made only when span_diff is there)
                    if (cp != 0) atomicAdd(a.cap_diff + hkey[i], cp);
                    hkey[i] = KEY_INVALID; hval[i] = 0;
                }
            __syncthreads();                              // the hash is empty before the next batch marks
        }
    }
    for (int o = 32; o > 0; o >>= 1) n_ev += __shfl_down(n_ev, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && n_ev) atomicAdd(&s_ev, n_ev);
    __syncthreads();
    if (threadIdx.x == 0 && s_ev) atomicAdd(a.n_ev, s_ev);
    if (__syncthreads_or(unphased ? 1 : 0) && threadIdx.x == 0) atomicOr(a.bad, 4u);
}

// The scatter of the (segment, tile) entries into their tiles' regions, with the atomics aggregated per workgroup in an LDS hash.  The
// segments of a coordinate-sorted BAM arrive gene by gene, so consecutive batches of 256 segments hit the same few tiles: a workgroup
// dequeues BIN_SUPER consecutive batches and keeps accumulating (batch, 8-tile round) items in the hash until it is 5/8 full, then
// claims a range per distinct tile with ONE global atomic for the whole chunk, replays the chunk's items and writes the entries'
// records.  (A deep gene funnels thousands of batches into a few cache lines of the cursors; same-line atomics serialise in L2 at
// ~90 per microsecond, so their number is what counts.)
constexpr int BIN_THREADS = 256;
constexpr int BIN_TPR = 8;             // tiles per segment handled per item
constexpr int BIN_H = 4096;            // LDS hash slots
constexpr int BIN_SUPER = 16;          // batches per dequeue
constexpr int BIN_MAXI = 32;           // items per chunk
constexpr uint32_t BIN_FILL = BIN_H * 5 / 8;

struct BinSeg { uint32_t key, tb, t0, rd; int32_t st, ln, ntile; int64_t evoff, b1; };      // ntile: ENTRIES of the segment = tiles it touches + window edges that cut it inside a tile; b1: the first window edge after its start

__device__ __forceinline__ BinSeg bin_load(const BuildArgs& a, int64_t s) {
    BinSeg g; g.key = KEY_INVALID; g.tb = 0; g.t0 = 0; g.rd = 0; g.b1 = 0; g.st = 0; g.ln = 0; g.ntile = 0; g.evoff = 0;
    // (all five loads at once, whatever the segment's admission says: a segment past the end reads the last one)
    const int64_t si = s < a.n_segs ? s : a.n_segs - 1;
    const uint2 info = a.seg_info[si];
    const int64_t st = a.seg_start[si], ln = a.seg_len[si], evoff = a.seg_ev_off[si];
    const uint32_t rd = a.seg_read[si];
    const int bs = 6 + a.wsh;                            // positions per bin: 1 << bs (a contig's first tile is even: its bins are its positions >> bs)
    if (s < a.n_segs && info.x != KEY_INVALID) {
        g.key = info.x; g.tb = info.y >> a.wsh;
        g.st = (int32_t)st; g.ln = (int32_t)ln; g.evoff = evoff; g.rd = rd;
        g.t0 = g.tb + ((uint32_t)g.st >> bs);
        g.ntile = (int)(((uint32_t)(g.st + g.ln - 1) >> bs) - ((uint32_t)g.st >> bs)) + 1;
        g.b1 = win_edge_after(g.st, a.window);
        for (int64_t b = g.b1; b < (int64_t)g.st + g.ln; b += a.window) g.ntile += (b & (int64_t)((1 << bs) - 1)) != 0;
    } else if (s < a.n_segs) g.tb = info.y >> a.wsh;
    return g;
}
// entry k of a segment, in position order: its tile (relative to the segment's first) and its positions [lo, hi)
__device__ __forceinline__ void bin_piece(const BinSeg& g, int32_t W, int bs, int k, uint32_t& tile_rel, int32_t& lo, int32_t& hi) {
    const int32_t en = g.st + g.ln;
    const int64_t bmask = (int64_t)((1 << bs) - 1);
    int shift = 0;
    int64_t b = g.b1;                                        // (almost always past the segment: no edge, no loop turn, no division)
    for (; b < en; b += W) {
        if ((b & bmask) == 0) continue;
        const int idx = (int)((b >> bs) - (g.st >> bs)) + shift + 1;           // the entry that STARTS at this edge
        if (k < idx) break;
        if (k == idx) {
            tile_rel = (uint32_t)((b >> bs) - (g.st >> bs)); lo = (int32_t)b;
            const int32_t tend = (int32_t)(((b >> bs) + 1) << bs);
            hi = en < tend ? en : tend;
            return;
        }
        ++shift;
    }
    tile_rel = (uint32_t)(k - shift);
    const int32_t tstart = (int32_t)((((uint32_t)g.st >> bs) + tile_rel) << bs);
    lo = g.st > tstart ? g.st : tstart;
    hi = en < tstart + (1 << bs) ? en : tstart + (1 << bs);
    if (b < hi && b > lo && (b & bmask) != 0) hi = (int32_t)b;   // the next edge (the one the loop stopped at) inside the rest of the bin ends the entry
}

__global__ __launch_bounds__(BIN_THREADS) void k_bin(BuildArgs a) {
    __shared__ uint32_t hkey[BIN_H], hcnt[BIN_H];      // hcnt turns into the tile's write cursor after the claim
    __shared__ uint32_t s_newb[BIN_MAXI], s_ib[BIN_MAXI], s_ir[BIN_MAXI];
    __shared__ int s_maxb[BIN_MAXI];
    __shared__ uint32_t s_super;
    const int t = threadIdx.x, lane = t & 63;
    constexpr int HSHIFT = 32 - __builtin_ctz(BIN_H);
    for (int i = t; i < BIN_H; i += BIN_THREADS) { hkey[i] = KEY_INVALID; hcnt[i] = 0; }
    const int64_t n_batches = (a.n_segs + BIN_THREADS - 1) / BIN_THREADS;
    const int64_t n_super = (n_batches + BIN_SUPER - 1) / BIN_SUPER;
    for (bool first = true;; first = false) {
        __syncthreads();
        // every workgroup's first item is its own index: no storm of same-address atomics at launch
        if (t == 0) s_super = first ? blockIdx.x : (uint32_t)atomicAdd(a.qhead, 1ull) + gridDim.x;
        __syncthreads();
        const int64_t sup = s_super;
        if (sup >= n_super) break;
        const int64_t b0 = sup * BIN_SUPER;
        const int64_t b1 = b0 + BIN_SUPER < n_batches ? b0 + BIN_SUPER : n_batches;
        int64_t cb = b0; int cr = 0;                 // next item: round cr of batch cb
        while (cb < b1) {
            // ---- pass A: accumulate items in the hash
            if (t < BIN_MAXI) { s_newb[t] = 0; s_maxb[t] = 0; }
            __syncthreads();
            int ni = 0; uint32_t tot = 0;
            int64_t b = cb; int r = cr;
            bool stop = false;
            while (!stop && b < b1) {
                const BinSeg g = bin_load(a, b * BIN_THREADS + t);
                const int ni_first = ni;
                int wmax = g.ntile;
                for (int o = 32; o > 0; o >>= 1) { int v = __shfl_down(wmax, o); wmax = v > wmax ? v : wmax; }
                if (lane == 0 && wmax) atomicMax(&s_maxb[ni_first], wmax);
                int R = -1;
                for (;;) {
                    uint32_t newc = 0;
#pragma unroll
                    for (int j = 0; j < BIN_TPR; ++j) {
                        const int k = r * BIN_TPR + j;
                        if (k < g.ntile) {
                            uint32_t trel; int32_t lo_, hi_;
                            bin_piece(g, a.window, 6 + a.wsh, k, trel, lo_, hi_);
                            const uint32_t x = g.t0 + trel;
                            uint32_t h = (x * 2654435761u) >> HSHIFT;
                            while (true) {
                                uint32_t prev = atomicCAS(&hkey[h], KEY_INVALID, x);
                                if (prev == KEY_INVALID) { ++newc; break; }
                                if (prev == x) break;
                                h = (h + 1) & (BIN_H - 1);
                            }
                            atomicAdd(&hcnt[h], 1u);
                        }
                    }
                    for (int o = 32; o > 0; o >>= 1) newc += __shfl_down(newc, o);
                    if (lane == 0 && newc) atomicAdd(&s_newb[ni], newc);
                    if (t == 0) { s_ib[ni] = (uint32_t)(b - b0); s_ir[ni] = (uint32_t)r; }
                    __syncthreads();
                    if (R < 0) R = (s_maxb[ni_first] + BIN_TPR - 1) / BIN_TPR;
                    tot += s_newb[ni];
                    ++ni; ++r;
                    if (r >= R) { ++b; r = 0; }
                    if (ni >= BIN_MAXI || tot + BIN_THREADS * BIN_TPR > BIN_FILL) { stop = true; break; }
                    if (r == 0) break;
                }
            }
            // ---- one global atomic per distinct tile of the chunk
            for (int i = t; i < BIN_H; i += BIN_THREADS) {
                const uint32_t cnt = hcnt[i];
                if (cnt) hcnt[i] = atomicAdd(&a.cursor[hkey[i]], cnt);       // first place of this workgroup's range in the tile's region
            }
            __syncthreads();
            {
                // ---- pass B: replay the items, write the entries
                for (int it = 0; it < ni; ++it) {
                    const int64_t bb = b0 + s_ib[it];
                    const int rr = (int)s_ir[it];
                    const BinSeg g = bin_load(a, bb * BIN_THREADS + t);
#pragma unroll
                    for (int j = 0; j < BIN_TPR; ++j) {
                        const int k = rr * BIN_TPR + j;
                        if (k < g.ntile) {
                            uint32_t trel; int32_t lo, hi;
                            bin_piece(g, a.window, 6 + a.wsh, k, trel, lo, hi);
                            const uint32_t x = g.t0 + trel;
                            uint32_t h = (x * 2654435761u) >> HSHIFT;
                            while (hkey[h] != x) h = (h + 1) & (BIN_H - 1);
                            const uint32_t pos = atomicAdd(&hcnt[h], 1u);
                            const int32_t tstart = (int32_t)((x - g.tb) << (6 + a.wsh));
                            // the entry lies in the window that STARTS inside its tile: an edge in (tstart, lo].  Edges the segment crosses are
                            // known (b1, b1 + W, ...); a segment that starts behind its tile's edge finds it one window before b1
                            int64_t edge = -1;
                            { int64_t e = g.b1 - a.window; while (e + a.window <= lo) e += a.window; if (e > tstart && e <= lo && e > 1) edge = e; }
                            const uint64_t src = (uint64_t)(g.evoff + (lo - g.st)) >> a.src_shift;      // the entry's first event in the caller's array (tile-phased events: its line)
                            // everything the gather needs of an entry travels THROUGH the sort (no record fetched through the sort's
                            // permutation afterwards): key = barcode | first position in the tile | events - 1 | source of the events,
                            // sorted on its barcode bits only; value = owning read | first of its segment | forward strand
                            const uint64_t key = sort_key(g.key & CB_MASK, (uint32_t)(lo - tstart), (uint32_t)(hi - lo - 1), src, a.cb_bits, a.wsh);
                            const uint32_t flags = (lo == g.st ? RV_SEGFIRST : 0u) | (((g.key >> 24) & 1u) ? 0u : RV_FWD);
                            if (a.rdv) { a.key[pos] = key; a.rdv[pos] = g.rd | flags | (edge >= 0 && lo >= edge ? RV_WHI : 0u); }
                            else a.key[pos] = key | ((uint64_t)flags << 32);      // keys alone (build_store, keys_only): the two flags a count needs above the source field, bits 62 and 63
                        }
                    }
                }
                __syncthreads();
                for (int i = t; i < BIN_H; i += BIN_THREADS)
                    if (hkey[i] != KEY_INVALID) { hkey[i] = KEY_INVALID; hcnt[i] = 0; }
            }
            cb = b; cr = r;
        }
    }
}

// ---- the tiles' tables in five launches ---------------------------------------------------------------
// What follows from the range marks k_seg_static left: entries per tile (running sum of cap_diff), their offsets, the largest running
// sum of the span marks (the depth cap's all-reads bound), the non-empty tiles in order, blocks per tile and their offsets, the 64-bit
// total.  As library calls that was five scans, five reductions, a select and a transform over all tiles of the genome - some twenty
// launches, 0.26 ms whatever the load's size (a rank's eighth of a sample pays it in full).  Here: every workgroup owns a contiguous
// chunk of tiles; k_tt_marks sums its marks, k_tt_scan1 (one workgroup) turns the chunks' sums into their starting values,
// k_tt_caps writes the capacities and sums what derives from them, k_tt_scan2 again, k_tt_offsets writes the offsets and the list.
constexpr int TT_THREADS = 256, TT_SUB = TT_THREADS * 4, TT_MAX_CHUNKS = 1024;
struct TtAgg1 { int32_t d, s, smax, pad; };
struct TtAgg2 { unsigned long long c; uint32_t b, ne; };
struct TtArgs {
    const int32_t* cap_diff; const int32_t* span_diff; uint32_t T; uint32_t chunk;       // chunk: tiles per workgroup, a multiple of TT_SUB; indices 0 .. T
    uint32_t n_chunks;
    TtAgg1* agg1; int32_t* din; TtAgg2* agg2; TtAgg2* in2;
    uint32_t* cap; uint32_t* tile_off; uint32_t* blk_off; uint32_t* netile;
    uint32_t* d_small; unsigned long long* d_sum;          // [1] non-empty tiles, [3] the live-reads bound; the 64-bit total
};
// exclusive prefix of v over the workgroup's threads (NT of them) and the total; two barriers
template <int NT, class V>
__device__ __forceinline__ V tt_wg_exscan(V v, V& total, V* sh /* NT / 64 + 1 */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    V inc = v;
    for (int o = 1; o < 64; o <<= 1) { const V u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    __syncthreads();                                   // (sh may still be read from the call before)
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    V base = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) { const V x = sh[w]; if (w < wv) base += x; tot += x; }
    total = tot;
    return base + inc - v;
}
__global__ __launch_bounds__(TT_THREADS) void k_tt_marks(TtArgs a) {
    __shared__ int32_t sh[TT_THREADS / 64 + 1];
    __shared__ int32_t sh_max[TT_THREADS / 64];
    const uint32_t lo = blockIdx.x * a.chunk, hi = lo + a.chunk < a.T + 1 ? lo + a.chunk : a.T + 1;
    int32_t dsum = 0, srun = 0, smax = INT32_MIN;
    for (uint32_t b0 = lo; b0 < hi; b0 += TT_SUB) {
        const uint32_t i0 = b0 + threadIdx.x * 4;
        int32_t d[4], sp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const bool in = i0 + k < hi; d[k] = in ? a.cap_diff[i0 + k] : 0; sp[k] = in && a.span_diff ? a.span_diff[i0 + k] : 0; }
        int32_t ds = d[0] + d[1] + d[2] + d[3];
        // the largest running sum of the span marks inside the thread's four, relative to the thread's start
        int32_t r = 0, m = INT32_MIN;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r += sp[k]; if (i0 + k < hi) m = r > m ? r : m; }
        int32_t tot_s; const int32_t ex = tt_wg_exscan<TT_THREADS>(r, tot_s, sh);
        if (m != INT32_MIN) { const int32_t v = srun + ex + m; smax = v > smax ? v : smax; }
        srun += tot_s; dsum += ds;
    }
    for (int o = 32; o > 0; o >>= 1) { const int32_t u = __shfl_down(smax, o); smax = u > smax ? u : smax; dsum += __shfl_down(dsum, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh_max[threadIdx.x >> 6] = smax; sh[threadIdx.x >> 6] = dsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t m = INT32_MIN, dd = 0;
        for (int w = 0; w < TT_THREADS / 64; ++w) { m = sh_max[w] > m ? sh_max[w] : m; dd += sh[w]; }
        a.agg1[blockIdx.x] = TtAgg1{dd, srun, m, 0};
    }
}
__global__ __launch_bounds__(TT_MAX_CHUNKS) void k_tt_scan1(TtArgs a) {
    __shared__ int32_t sh[TT_MAX_CHUNKS / 64 + 1];
    __shared__ int32_t sh_max[TT_MAX_CHUNKS / 64];
    const uint32_t i = threadIdx.x;
    TtAgg1 g{0, 0, INT32_MIN, 0};
    if (i < a.n_chunks) g = a.agg1[i];
    int32_t tot;
    const int32_t din = tt_wg_exscan<TT_MAX_CHUNKS>(g.d, tot, sh);
    const int32_t sin = tt_wg_exscan<TT_MAX_CHUNKS>(g.s, tot, sh);
    if (i < a.n_chunks) a.din[i] = din;
    int32_t m = i < a.n_chunks && g.smax != INT32_MIN ? sin + g.smax : INT32_MIN;
    for (int o = 32; o > 0; o >>= 1) { const int32_t u = __shfl_down(m, o); m = u > m ? u : m; }
    __syncthreads();
    if ((i & 63) == 0) sh_max[i >> 6] = m;
    __syncthreads();
    if (i == 0) { int32_t mm = INT32_MIN; for (int w = 0; w < TT_MAX_CHUNKS / 64; ++w) mm = sh_max[w] > mm ? sh_max[w] : mm; a.d_small[3] = (uint32_t)mm; }
}
__global__ __launch_bounds__(TT_THREADS) void k_tt_caps(TtArgs a) {
    __shared__ int32_t sh[TT_THREADS / 64 + 1];
    __shared__ unsigned long long sh_c[TT_THREADS / 64];
    __shared__ uint32_t sh_b[TT_THREADS / 64], sh_n[TT_THREADS / 64];
    const uint32_t lo = blockIdx.x * a.chunk, hi = lo + a.chunk < a.T + 1 ? lo + a.chunk : a.T + 1;
    int32_t run = a.din[blockIdx.x];
    unsigned long long csum = 0; uint32_t bsum = 0, nsum = 0;
    for (uint32_t b0 = lo; b0 < hi; b0 += TT_SUB) {
        const uint32_t i0 = b0 + threadIdx.x * 4;
        int32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = i0 + k < hi ? a.cap_diff[i0 + k] : 0;
        int32_t tot; const int32_t ex = tt_wg_exscan<TT_THREADS>(d[0] + d[1] + d[2] + d[3], tot, sh);
        int32_t r = run + ex;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r += d[k];
            if (i0 + k < hi) {
                a.cap[i0 + k] = (uint32_t)r;
                if (i0 + k < a.T) { const uint32_t cv = (uint32_t)r; csum += cv; bsum += (cv + 7u) / 8u; nsum += cv != 0u; }
            }
        }
        run += tot;
    }
    for (int o = 32; o > 0; o >>= 1) { csum += __shfl_down(csum, o); bsum += __shfl_down(bsum, o); nsum += __shfl_down(nsum, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh_c[threadIdx.x >> 6] = csum; sh_b[threadIdx.x >> 6] = bsum; sh_n[threadIdx.x >> 6] = nsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        TtAgg2 g{0, 0, 0};
        for (int w = 0; w < TT_THREADS / 64; ++w) { g.c += sh_c[w]; g.b += sh_b[w]; g.ne += sh_n[w]; }
        a.agg2[blockIdx.x] = g;
    }
}
__global__ __launch_bounds__(TT_MAX_CHUNKS) void k_tt_scan2(TtArgs a) {
    __shared__ unsigned long long sh64[TT_MAX_CHUNKS / 64 + 1];
    __shared__ uint32_t sh32[TT_MAX_CHUNKS / 64 + 1];
    const uint32_t i = threadIdx.x;
    TtAgg2 g{0, 0, 0};
    if (i < a.n_chunks) g = a.agg2[i];
    unsigned long long tc; uint32_t tb, tn;
    TtAgg2 in;
    in.c = tt_wg_exscan<TT_MAX_CHUNKS>(g.c, tc, sh64);
    in.b = tt_wg_exscan<TT_MAX_CHUNKS>(g.b, tb, sh32);
    in.ne = tt_wg_exscan<TT_MAX_CHUNKS>(g.ne, tn, sh32);
    if (i < a.n_chunks) a.in2[i] = in;
    if (i == 0) { *a.d_sum = tc; a.d_small[1] = tn; }
}
__global__ __launch_bounds__(TT_THREADS) void k_tt_offsets(TtArgs a) {
    __shared__ uint32_t sh[TT_THREADS / 64 + 1];
    const uint32_t lo = blockIdx.x * a.chunk, hi = lo + a.chunk < a.T + 1 ? lo + a.chunk : a.T + 1;
    const TtAgg2 in = a.in2[blockIdx.x];
    uint32_t crun = (uint32_t)in.c, brun = in.b, nrun = in.ne;
    for (uint32_t b0 = lo; b0 < hi; b0 += TT_SUB) {
        const uint32_t i0 = b0 + threadIdx.x * 4;
        uint32_t cv[4], bv[4], nv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cv[k] = i0 + k < hi && i0 + k < a.T ? a.cap[i0 + k] : 0u; bv[k] = (cv[k] + 7u) / 8u; nv[k] = cv[k] != 0u; }
        uint32_t tc, tb, tn;
        uint32_t ec = crun + tt_wg_exscan<TT_THREADS>(cv[0] + cv[1] + cv[2] + cv[3], tc, sh);
        uint32_t eb = brun + tt_wg_exscan<TT_THREADS>(bv[0] + bv[1] + bv[2] + bv[3], tb, sh);
        uint32_t en = nrun + tt_wg_exscan<TT_THREADS>(nv[0] + nv[1] + nv[2] + nv[3], tn, sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < hi) {
                a.tile_off[i0 + k] = ec; a.blk_off[i0 + k] = eb;
                if (nv[k]) a.netile[en] = i0 + k;
            }
            ec += cv[k]; eb += bv[k]; en += nv[k];
        }
        crun += tc; brun += tb; nrun += tn;
    }
}
// per tile: blocks, and a flag for the non-empty ones
__global__ void k_tile_blocks(const uint32_t* cap, uint32_t n_tiles, uint32_t* blk) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n_tiles) blk[t] = t < n_tiles ? (cap[t] + 7u) / 8u : 0u;
}
struct CapNonZero {
    const uint32_t* cap;
    __host__ __device__ bool operator()(const uint32_t& t) const { return cap[t] != 0; }
};
// split: the segments of more than SORT_LARGE entries in (begin, end), the others in (begin2, end2) - each list names every tile, a tile of
// the other class as an empty segment (the two sorts run side by side on two streams; an empty segment costs a sort a few lanes)
constexpr uint32_t SORT_LARGE = 256;      // rocprim's block-per-segment kernel takes the segments beyond the warp sorts' 32 x 8 items (LsgSortConfig)
__global__ void k_seg_bounds(const uint32_t* netile, uint32_t n, const uint32_t* tile_off, uint32_t* begin, uint32_t* end, uint32_t* begin2, uint32_t* end2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = netile[i], b = tile_off[t], e = tile_off[t + 1];
    if (!begin2) { begin[i] = b; end[i] = e; return; }
    const bool large = e - b > SORT_LARGE;
    begin[i] = b; end[i] = large ? e : b;
    begin2[i] = b; end2[i] = large ? b : e;
}
__global__ void k_tile_caps(const uint32_t* tiles, uint32_t n, const uint32_t* cap, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cap[tiles[i]] >> 11;
}
// the tile of every block: a non-empty tile writes its number at its first block, a running maximum over the blocks carries it on
// (the tiles' numbers grow with the blocks; a binary search per block over the tiles' offsets took 0.27 ms at C2)
__global__ void k_tm_blk_mark(const uint32_t* cap, const uint32_t* blk_off, uint32_t n_tiles, uint32_t* blk_tile) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles && cap[t]) blk_tile[blk_off[t]] = t;
}

// One wave per TMG_BLOCKS blocks = 8 TMG_BLOCKS entries, in two steps.
// (a) Lanes 0 .. 8 TMG_BLOCKS - 1, one entry each, in sorted order: the entry's packed key and value as the sort left them (read in
//     place: nothing is fetched through a permutation), its run flags (neighbouring keys' barcodes) -> the store's per-entry words s0, b,
//     rd; where its events lie stays in the lane's registers.
// (b) Lane = (entry u of a block, 16-byte chunk c of its <= 128 bytes): ONE load instruction per block fetches all eight entries from
//     wherever they lie in the caller's array (2-byte aligned: the hardware takes unaligned dwordx4).  The chunks cross an LDS tile
//     [entry][64 events]; lane = position then picks, per entry, the event at (position - first position of the entry), and the rows
//     between the block's first and last position with an event leave as one transposed kilobyte (rows outside that extent are never
//     read: the walk's buffer descriptor ends there, genotype.hip tests it).
constexpr int TMG_BLOCKS = 4, TMG_WAVES = 4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U4A2 { u32x4 v; };
__global__ __launch_bounds__(TMG_WAVES * 64) void k_tm_gather(const uint16_t* events, int64_t n_events, const uint64_t* key, const uint32_t* rdv, int cb_bits, int src_shift,
                                                               const uint32_t* tile_off, const uint32_t* blk_off, const uint32_t* blk_tile, uint32_t nblk,
                                                               uint32_t* s0, uint8_t* b8, uint32_t* rd, uint4* store, uint16_t* ext) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[TMG_WAVES][TMG_BLOCKS][8][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin, each with its own L2.  An entry's <= 128 source bytes sit at a 2-byte alignment,
    // so the 128-byte line they end in is also the line the SAME read's entry of the NEXT tile starts in: give every XCD one contiguous
    // eighth of the blocks, so that neighbouring tiles pass through one L2 shortly after one another (measured: 54 GB of fabric reads
    // per launch for 17.6 GB of events with the plain mapping, 37 GB with this one).  A speed heuristic only: nothing depends on it.
    // (Tried on top, round 3: 16 neighbouring tiles advancing together through their barcode order — the units taken in the order of
    // (tile / 16, relative place in the tile) — because half of C2's entries sit in tiles deeper than 27 000 entries, where the two
    // uses of a shared line lie 10 MB apart.  Reads fell to 32 GB, the kernel's time did not move (9.6 ms) and the order costs a
    // 0.3 ms sort: the kernel runs at the 5.5 TB/s this chip copies at, whatever the mix of its 53 GB.  Not kept.)
    const uint32_t per_xcd = gridDim.x >> 3;                 // (the grid is a multiple of 8)
    const uint32_t vwg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t blk0 = (vwg * TMG_WAVES + (uint32_t)wv) * TMG_BLOCKS;
    if (blk0 >= nblk) return;
    // ---- (a)
    uint32_t e_src = 0, e_info = 0;                 // source of the entry's events, low 32 bits; high 8 bits [0..7] | first position [8..13] | events [16..22] (0: not there)
    if (lane < 8 * TMG_BLOCKS) {
        const uint32_t blk = blk0 + ((uint32_t)lane >> 3);
        if (blk < nblk) {
            const uint64_t p = (uint64_t)blk * 8 + (lane & 7);
            const uint32_t t = blk_tile[blk];
            const uint32_t i = (uint32_t)(p - (uint64_t)blk_off[t] * 8), off = tile_off[t], n = tile_off[t + 1] - off;
            // (everything this kernel writes is written once and read by another kernel: non-temporal stores keep it from pushing the
            // source lines that neighbouring tiles share out of L2 — 10.4 -> 9.6 ms)
            auto put = [](auto* q, auto v) { __builtin_nontemporal_store(v, q); };
            if (i >= n) { put(s0 + p, (uint32_t)TM_PAD_S0); put(b8 + p, (uint8_t)0); put(rd + p, 0u); }
            else {
                const uint32_t j = off + i, cbm = (1u << cb_bits) - 1u;
                const uint64_t k64 = key[j];
                const uint32_t k = (uint32_t)k64 & cbm, v = rdv[j];
                const bool rs = i == 0 || ((uint32_t)key[j - 1] & cbm) != k;
                const bool single = rs && (i + 1 == n || ((uint32_t)key[j + 1] & cbm) != k);
                const uint32_t geom = (uint32_t)(k64 >> cb_bits), first = geom & 63u, nev1 = (geom >> 6) & 63u;
                const uint64_t src = ((k64 >> (cb_bits + 12)) << src_shift) | (src_shift ? first : 0u);
                put(s0 + p, k | ((v & RV_FWD) ? TM_FWD : 0u) | ((v & RV_WHI) ? TM_WHI : 0u) | (rs ? TM_RUNSTART : 0u));
                put(b8 + p, (uint8_t)(nev1 | ((v & RV_SEGFIRST) ? 64u : 0u) | (single ? 128u : 0u)));
                put(rd + p, v & RV_READ);
                e_src = (uint32_t)src; e_info = (uint32_t)(src >> 32) | (first << 8) | ((nev1 + 1u) << 16);
            }
        }
    }
    // ---- (b)
    const int u = lane >> 3, c = lane & 7;
    u32x4 chunk[TMG_BLOCKS];
    uint32_t info[TMG_BLOCKS];
#pragma unroll
    for (int q = 0; q < TMG_BLOCKS; ++q) {
        chunk[q] = u32x4{0u, 0u, 0u, 0u};
        const uint32_t lo32 = (uint32_t)__shfl((int)e_src, q * 8 + u), inf = (uint32_t)__shfl((int)e_info, q * 8 + u);
        info[q] = inf;
        const uint32_t nev = inf >> 16;
        if ((uint32_t)c * 8u < nev) {
            const int64_t off = (int64_t)(((uint64_t)(inf & 0xffu) << 32) | lo32) + c * 8;
            if (off + 8 <= n_events) chunk[q] = reinterpret_cast<const U4A2*>(events + off)->v;
            else {                                   // the last events of the array: never read past it
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int i = 0; i < 8; ++i) if (off + i < n_events) w[i >> 1] |= (uint32_t)events[off + i] << (16 * (i & 1));
                chunk[q] = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TMG_BLOCKS; ++q) *reinterpret_cast<u32x4*>(&lds[wv][q][u][c * 8]) = chunk[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < TMG_BLOCKS; ++q) {
        const uint32_t blk = blk0 + q;
        if (blk >= nblk) break;
        uint32_t e[8], any = 0;
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) {
            const uint32_t inf = (uint32_t)__builtin_amdgcn_readlane((int)info[q], uu * 8);
            const uint32_t idx = (uint32_t)lane - ((inf >> 8) & 63u);
            e[uu] = idx < (inf >> 16) ? (uint32_t)lds[wv][q][uu][idx & 63u] : 0u;
            any |= e[uu];
        }
        const unsigned long long m = __ballot(any != 0u);
        const int first = m ? __ffsll((long long)m) - 1 : 0, last = m ? 64 - __clzll((long long)m) : 0;
        if (lane >= first && lane < last) {
            const u32x4 row = {e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
            u32x4* dst = reinterpret_cast<u32x4*>(store) + (uint64_t)blk * 64 + lane;
            __builtin_nontemporal_store(row, dst);
        }
        if (lane == 0) ext[blk] = (uint16_t)(first | (last << 8));
    }
}

#define SCAN_U32(in, out, n)                                                                              \
    do {                                                                                                  \
        size_t tb_ = 0;                                                                                   \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_, (in), (out), (int)(n), st));              \
        if (c->d_cub_tmp.reserve(tb_ + 256)) return -1;                                                   \
        tb_ = c->d_cub_tmp.cap;                                                                           \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb_, (in), (out), (int)(n), st));       \
    } while (0)

enum { BT_KEY_A = 0, BT_KEY_B, BT_VAL_A, BT_VAL_B, BT_PEX, BT_NETILE, BT_SEG_BEGIN, BT_SEG_END, BT_TMP, BT_PER_TILE, BT_OFFS, BT_SPAN, BT_SPAN_RUN, BT_BLK, BT_CURSOR, BT_LPT, BT_COPY_TMP, BT_N };

void drop_store(lsg_ctx* c) {
    c->tm_valid = false; c->plan_n_ct = 0; c->plan1_n_ct = 0; c->tm_n = 0; c->tm_events = 0; c->tm_np = 0; c->tm_nblk = 0; c->tm_njobs = 0; c->tm_nchunks = 0;
    c->tm_n_ne = 0; c->tm_n_multi = 0; c->tm_n_slabs = 0; c->tm_n_wide = 0;
    c->max_live_reads = -1; c->max_live_all = -1; c->max_live_exact = -1; c->has_drops = false;
    c->counted = c->called = false; c->counted_at_load = false; c->load_was_fused = false; c->store_skipped = false;
}

// the build's temporaries stay in the context (grow-only): allocating ~9 GB per load costs more wall time than the build's kernels —
// unless the load is large against the device (C4): then they are given back, the rows and the call stage need the room
static void settle_temporaries(lsg_ctx* c) {
    size_t held = 0, mem_free = 0, mem_total = 0;
    for (auto& b : c->bt) held += b.cap;
    held += c->ws[WS_SEG_INFO].cap;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return;
    // (a load that kept no store holds little else: its temporaries stay while a quarter of the device is free - giving 46 GB back and
    // asking for them again cost C4's next load 3.8 s)
    const bool tight = c->store_skipped ? mem_free < mem_total / 4 : (held > mem_total / 8 || mem_free < mem_total / 8);
    if (tight) { for (auto& b : c->bt) b.release(); c->ws[WS_SEG_INFO].release(); c->plan1_n_ct = 0; }      // (the plan's tile-level half lived there: the first count makes it again)
}

static int plan_tiles(lsg_ctx* c, hipStream_t st);
static int plan_jobs(lsg_ctx* c, bool finish, const uint64_t* skey = nullptr, int cb_bits = 0);
static void plan_finish(lsg_ctx* c);
static void plan_finish_tiles(lsg_ctx* c);

// rocprim's segmented radix sort with a configuration of its own.  gfx950 gets rocprim's generic default (3.7 ms for C2's 185 M pairs in
// 5 * 10^5 segments); measured on the MI355X over radix bits 6-8, blocks of 256 / 512 threads with 4-24 items per thread and four
// warp-sort shapes: 7 bits per pass (13-bit barcodes: 7 + 6), 256 x 8 items in the block sort, segments of up to 64 / 256 pairs to the
// small / medium warp sorts: 2.3 ms.  Where 8 bits per pass save a pass (C4's 20 000 barcodes: 15 key bits in two passes) they are taken.
template <unsigned RB, unsigned BS = 256, unsigned IPT = 8>
using LsgSortConfig = rocprim::segmented_radix_sort_config<RB, rocprim::kernel_config<BS, IPT>, rocprim::WarpSortConfig<16, 4, 256, 64, 32, 8, 256>>;
template <unsigned RB, unsigned BS, unsigned IPT, class... Args>
static hipError_t lsg_segmented_sort(Args&&... args) { return rocprim::segmented_radix_sort_pairs<LsgSortConfig<RB, BS, IPT>>(std::forward<Args>(args)...); }
template <unsigned RB, unsigned BS, unsigned IPT, class... Args>
static hipError_t lsg_segmented_sort_keys(Args&&... args) { return rocprim::segmented_radix_sort_keys<LsgSortConfig<RB, BS, IPT>>(std::forward<Args>(args)...); }

int build_store(lsg_ctx* c, const uint16_t* events, int64_t n_events, const int64_t* seg_ev_off, const lsg_reads* src) {
    drop_store(c);
    hipStream_t st = c->stream;
    const int64_t S = c->rd.n_segs, R = c->rd.n_reads;
    // WINDOWS.  A load that will keep no store and sort keys alone (below), over events its producer says are phased modulo 128
    // (lsg_set_events_layout; this library's own producers say so themselves), bins its entries by 128-position windows - tiles (2 w, 2 w + 1) -
    // instead of tiles: 0.58 x as many entries through the scatter and the sort, and every entry one aligned 256-byte block for the count
    // (k_tm_count_win).  Everything here that is "per tile" is then per window (T bins); the count's units stay the tiles'.  Should the early
    // look say otherwise (events not phased after all, a depth cap that could fire, ...), the load starts again by tiles (win_off).
    const bool win = c->cal_enabled && c->store_policy == LSG_STORE_SKIP_WHEN_COUNTED && !getenv("LSG_NO_DIRECT_COUNT") && !getenv("LSG_NO_FUSED_LOAD") && !getenv("LSG_NO_KEYS_ONLY") &&
                     !getenv("LSG_NO_WINDOWS") && !c->keys_only_off && !c->win_off && c->n_ct >= 1 && c->n_ct <= 2 && c->n_cb > 0 && c->copy_stream && n_events >= 128 && n_events < (1ll << 39) &&
                     (c->events_layout == LSG_LAYOUT_PHASED || (c->hint_phased_events && c->hint_phased_events == (const void*)events)) && ((uintptr_t)events & 255u) == 0 && c->plp_window >= 128 &&
                     c->cal_params.min_bq >= 1 && c->cal_params.min_bq <= 255 && (c->n_tiles & 1u) == 0;
    const int wsh = win ? 1 : 0;
    c->wsh = wsh;
    auto start_again_by_tiles = [&]() -> int {
        LSG_HIP(hipStreamSynchronize(st)); LSG_HIP(hipStreamSynchronize(c->copy_stream));
        if (getenv("LSG_TIMING")) fprintf(stderr, "[lsg] load: not a load for 128-position windows after all, starting again by tiles\n");
        c->win_off = true;
        const int rc = build_store(c, events, n_events, seg_ev_off, src);
        c->win_off = false;
        return rc;
    };
    const uint32_t T = c->n_tiles >> wsh;
    const auto t_wall = std::chrono::steady_clock::now();
    auto finish = [&]() {
        c->layout_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
        c->tm_valid = true;
        return 0;
    };
    for (auto& v : c->build_ms) v = 0;
    c->st_min_mq = c->lf_min_mq; c->st_flag_exclude = c->lf_flag_exclude; c->st_ignore_orphans = c->lf_ignore_orphans;
    if (c->d_tile_cap.reserve(((size_t)T + 2) * 4) || c->d_tile_off.reserve(((size_t)T + 2) * 4) || c->d_scalars.reserve(512 * 8)) return -1;
    LSG_HIP(hipMemsetAsync(c->d_tile_cap.p, 0, ((size_t)T + 2) * 4, st));
    LSG_HIP(hipMemsetAsync(c->d_tile_off.p, 0, ((size_t)T + 2) * 4, st));
    if (S <= 0 || T == 0) { LSG_HIP(hipStreamSynchronize(st)); return finish(); }
    if (c->ws[WS_SEG_INFO].reserve(((size_t)S + 1) * 8)) return -1;
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, 64, st));
    BuildArgs a{};
    a.n_reads = R; a.n_segs = S; a.n_events = n_events;
    const lsg_reads& in = src ? *src : c->rd;          // (same contents: the handle's copies of a caller's device arrays may still be travelling on the copy stream)
    a.read_tid = in.read_tid; a.read_flag = in.read_flag; a.read_mapq = in.read_mapq; a.read_cb = in.read_cb;
    a.seg_read = in.seg_read; a.seg_start = in.seg_start; a.seg_len = in.seg_len; a.seg_ev_off = seg_ev_off;
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.contig_len = c->d_contig_len.as<int64_t>(); a.n_contigs = c->n_contigs; a.n_tiles = T;
    a.seg_info = c->ws[WS_SEG_INFO].as<uint2>(); a.tile_cap = c->d_tile_cap.as<uint32_t>();
    a.lf_min_mq = c->lf_min_mq; a.lf_flag_exclude = c->lf_flag_exclude; a.lf_ignore_orphans = c->lf_ignore_orphans;
    a.window = c->plp_window; c->st_window = c->plp_window; a.wsh = wsh;
    a.qhead = c->d_scalars.as<unsigned long long>(); a.bad = reinterpret_cast<uint32_t*>(c->d_scalars.as<unsigned long long>() + 2);
    a.n_ev = c->d_scalars.as<unsigned long long>() + 3;
    if (c->bt[BT_SPAN].reserve(((size_t)T + 2) * 4) || c->bt[BT_SPAN_RUN].reserve(((size_t)T + 2) * 4)) return -1;
    LSG_HIP(hipMemsetAsync(c->bt[BT_SPAN].p, 0, ((size_t)T + 2) * 4, st));
    a.span_diff = c->bt[BT_SPAN].as<int32_t>();
    uint32_t* d_small = reinterpret_cast<uint32_t*>(c->d_scalars.as<unsigned long long>() + 4);      // [0] max entries of a tile, [1] non-empty tiles, [2] largest barcode id, [3] the live-reads bound
    DevBuf& tmp = c->bt[BT_TMP];
    LSG_HIP(hipEventRecord(c->evb[0], st));
    // ---- 1. static admission + capacities
    unsigned g_seg = (unsigned)((S + 256 * SEG_U - 1) / (256 * SEG_U)); if (g_seg > (unsigned)(c->n_cus * 16)) g_seg = (unsigned)(c->n_cus * 16);
    unsigned g_bin = (unsigned)((S + 256 * BIN_SUPER - 1) / (256 * BIN_SUPER)); if (g_bin > (unsigned)(c->n_cus * 8)) g_bin = (unsigned)(c->n_cus * 8);
    if (c->bt[BT_PER_TILE].reserve(((size_t)T + 2) * 4)) return -1;
    LSG_HIP(hipMemsetAsync(c->bt[BT_PER_TILE].p, 0, ((size_t)T + 2) * 4, st));
    a.cap_diff = c->bt[BT_PER_TILE].as<int32_t>();
    hipLaunchKernelGGL(k_seg_static, dim3(g_seg), dim3(256), 0, st, a);
    if (c->bt[BT_NETILE].reserve(((size_t)T + 2) * 4) || c->bt[BT_BLK].reserve(((size_t)T + 2) * 4) || c->tm[TM_BLK_OFF].reserve(((size_t)T + 2) * 4)) return -1;
    uint32_t* blk_off = c->tm[TM_BLK_OFF].as<uint32_t>();
    unsigned long long* d_sum = c->d_scalars.as<unsigned long long>() + 8;
    {
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceReduce::Max(nullptr, tb, in.read_cb, reinterpret_cast<int32_t*>(d_small + 2), (int)(R > 0 ? R : 1), st));
        if (tmp.reserve(tb + 256 + 64 * 1024)) return -1;
        tb = tmp.cap;
        if (R > 0) LSG_HIP(hipcub::DeviceReduce::Max(tmp.p, tb, in.read_cb, reinterpret_cast<int32_t*>(d_small + 2), (int)R, st));
    }
    if (!getenv("LSG_NO_TILE_TABLES")) {
        // everything that follows from the range marks, in five launches (k_tt_*): the sizes below reach the host in the load's ONE early
        // look (the scatter, the sort and the gather are then queued without waiting for one another)
        if (c->d_cub_tmp.reserve(64 * 1024)) return -1;
        TtArgs ta{};
        ta.cap_diff = a.cap_diff; ta.span_diff = a.span_diff; ta.T = T;
        const uint64_t per = ((uint64_t)T + 1 + TT_MAX_CHUNKS - 1) / TT_MAX_CHUNKS;
        ta.chunk = (uint32_t)((per + TT_SUB - 1) / TT_SUB * TT_SUB);
        ta.n_chunks = (uint32_t)(((uint64_t)T + 1 + ta.chunk - 1) / ta.chunk);
        char* scratch = reinterpret_cast<char*>(c->d_cub_tmp.p);          // (a buffer nothing else on this stream holds across these launches)
        ta.agg1 = reinterpret_cast<TtAgg1*>(scratch); ta.agg2 = reinterpret_cast<TtAgg2*>(scratch + 16 * 1024); ta.in2 = reinterpret_cast<TtAgg2*>(scratch + 32 * 1024);
        ta.din = reinterpret_cast<int32_t*>(scratch + 48 * 1024);
        ta.cap = c->d_tile_cap.as<uint32_t>(); ta.tile_off = c->d_tile_off.as<uint32_t>(); ta.blk_off = blk_off; ta.netile = c->bt[BT_NETILE].as<uint32_t>();
        ta.d_small = d_small; ta.d_sum = d_sum;
        hipLaunchKernelGGL(k_tt_marks, dim3(ta.n_chunks), dim3(TT_THREADS), 0, st, ta);
        hipLaunchKernelGGL(k_tt_scan1, dim3(1), dim3(TT_MAX_CHUNKS), 0, st, ta);
        hipLaunchKernelGGL(k_tt_caps, dim3(ta.n_chunks), dim3(TT_THREADS), 0, st, ta);
        hipLaunchKernelGGL(k_tt_scan2, dim3(1), dim3(TT_MAX_CHUNKS), 0, st, ta);
        hipLaunchKernelGGL(k_tt_offsets, dim3(ta.n_chunks), dim3(TT_THREADS), 0, st, ta);
    } else {
    {   // entries per tile = running sum of the segments' range marks
        size_t tb = 0;
        int32_t* cap = reinterpret_cast<int32_t*>(c->d_tile_cap.as<uint32_t>());
        LSG_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, a.cap_diff, cap, (int)(T + 1), st));
        if (c->d_cub_tmp.reserve(tb + 256)) return -1;
        tb = c->d_cub_tmp.cap;
        LSG_HIP(hipcub::DeviceScan::InclusiveSum(c->d_cub_tmp.p, tb, a.cap_diff, cap, (int)(T + 1), st));
    }
    SCAN_U32(c->d_tile_cap.as<uint32_t>(), c->d_tile_off.as<uint32_t>(), T + 1);
    {   // the depth cap's table-independent bound (layout.hip live_read_bound_all): reads of any cell type whose span touches a tile, maximum over tiles
        int32_t* run = c->bt[BT_SPAN_RUN].as<int32_t>();
        size_t tb = 0, tb2 = 0;
        LSG_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, a.span_diff, run, (int)(T + 1), st));
        LSG_HIP(hipcub::DeviceReduce::Max(nullptr, tb2, run, reinterpret_cast<int32_t*>(d_small + 3), (int)(T + 1), st));
        if (tmp.reserve((tb > tb2 ? tb : tb2) + 256)) return -1;
        tb = tb2 = tmp.cap;
        LSG_HIP(hipcub::DeviceScan::InclusiveSum(tmp.p, tb, a.span_diff, run, (int)(T + 1), st));
        LSG_HIP(hipcub::DeviceReduce::Max(tmp.p, tb2, run, reinterpret_cast<int32_t*>(d_small + 3), (int)(T + 1), st));
    }
    // the non-empty tiles (the sort's segments) and the blocks of every tile: all of it follows from the capacities, so the sizes below
    // reach the host in the load's ONE early look (the scatter, the sort and the gather are then queued without waiting for one another)
    {
        hipcub::CountingInputIterator<uint32_t> tile_it(0);
        CapNonZero pred{c->d_tile_cap.as<uint32_t>()};
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceSelect::If(nullptr, tb, tile_it, c->bt[BT_NETILE].as<uint32_t>(), d_small + 1, (int)T, pred, st));
        if (tmp.reserve(tb + 256)) return -1;
        tb = tmp.cap;
        LSG_HIP(hipcub::DeviceSelect::If(tmp.p, tb, tile_it, c->bt[BT_NETILE].as<uint32_t>(), d_small + 1, (int)T, pred, st));
    }
    hipLaunchKernelGGL(k_tile_blocks, dim3((T + 256) / 256), dim3(256), 0, st, c->d_tile_cap.as<uint32_t>(), T, c->bt[BT_BLK].as<uint32_t>());
    SCAN_U32(c->bt[BT_BLK].as<uint32_t>(), blk_off, T + 1);
    // (a total of 2^32 or more wraps the 32-bit scan: the per-tile capacities are summed in 64 bits to tell)
    {
        size_t tb = 0;
        hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, const uint32_t*> it(c->d_tile_cap.as<uint32_t>(), hipcub::CastOp<unsigned long long>());
        LSG_HIP(hipcub::DeviceReduce::Sum(nullptr, tb, it, d_sum, (int)T, st));
        if (tmp.reserve(tb + 256)) return -1;
        tb = tmp.cap;
        LSG_HIP(hipcub::DeviceReduce::Sum(tmp.p, tb, it, d_sum, (int)T, st));
    }
    }
    uint32_t total = 0, bad = 0, n_netile = 0, nblk = 0; int32_t max_cb = 0, max_live = 0;
    unsigned long long n_ev = 0, sum = 0;
    // the load's first look at the device: the sizes of everything that follows, in TWO small copies to pinned memory (the scalars'
    // block, and the number of blocks where the scan left it) - seven separate copies were seven commands of ~20 us each on the stream
    LSG_HIP(hipMemcpyAsync(c->h_pin, c->d_scalars.p, 9 * 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(c->h_pin + 16, blk_off + T, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    {
        const unsigned long long* hp = c->h_pin;
        const uint32_t* hs = reinterpret_cast<const uint32_t*>(hp + 4);      // d_small[0..3]
        bad = (uint32_t)hp[2]; n_ev = hp[3]; n_netile = hs[1]; max_cb = R > 0 ? (int32_t)hs[2] : 0; max_live = (int32_t)hs[3]; sum = hp[8];
        nblk = (uint32_t)hp[16];
    }
    // tile-phased events (LSG_LAYOUT_PHASED): every entry lies inside one aligned 128-byte line of the caller's array; the key carries the line
    // (k_seg_static looked at the phase modulo the bins' width: 64, or 128 for windows)
    if (win && (bad & 4u) && !(bad & 3u)) return start_again_by_tiles();
    const int src_shift = !(bad & 4u) && ((uintptr_t)events & (wsh ? 255u : 127u)) == 0 && !getenv("LSG_NO_PHASED") ? 6 + wsh : 0;
    c->src_phased = src_shift != 0;
    if (bad & 2u) { set_error("lsg_load_reads: a segment's read index lies outside the read arrays"); return -2; }
    if (bad & 1u) { set_error("lsg_load_reads: a segment's event range lies outside the events array"); return -2; }
    if (sum >= 0x7FFFFFF0ull) { set_error("lsg_load_reads: %llu tile entries exceed the 31-bit entry index; load the reads in windows", sum); return -2; }
    total = (uint32_t)sum;
    const uint64_t N = total;
    c->tm_n = N; c->tm_events = (int64_t)n_ev;
    c->max_live_all = max_live > 0 ? max_live : 0;
    if (N == 0) return finish();
    if (getenv("LSG_TILE_HIST")) {                                   // (diagnostic: how the entries spread over tile sizes; C2: DESIGN.md section 9)
        std::vector<uint32_t> caps(T);
        LSG_HIP(hipMemcpy(caps.data(), c->d_tile_cap.p, (size_t)T * 4, hipMemcpyDeviceToHost));
        unsigned long long nt[8] = {0}, ne[8] = {0}; uint32_t mx = 0;
        const uint32_t lim[7] = {64, 256, 2048, 8192, 32768, 131072, 524288};
        for (uint32_t i = 0; i < T; ++i) { const uint32_t v = caps[i]; if (!v) continue; int k = 0; while (k < 7 && v > lim[k]) ++k; ++nt[k]; ne[k] += v; if (v > mx) mx = v; }
        for (int k = 0; k < 8; ++k) fprintf(stderr, "tiles <= %u: %llu tiles, %llu entries\n", k < 7 ? lim[k] : 0xffffffffu, nt[k], ne[k]);
        fprintf(stderr, "largest tile: %u entries\n", mx);
    }
    // ---- 2. scatter (queued BEFORE the copy stream's work below: those two dozen launches are 0.3 ms of host time the scatter need not wait for)
    int bits = 1; while (bits < 24 && (1ll << bits) <= (long long)max_cb) ++bits;      // the barcode bits of the sort key
    DevBuf &key_a = c->bt[BT_KEY_A], &key_b = c->bt[BT_KEY_B], &val_a = c->bt[BT_VAL_A], &val_b = c->bt[BT_VAL_B];
    if ((n_events >> src_shift) >= (1ll << (52 - bits - 2 * wsh))) {
        set_error("lsg_load_reads: %lld events with %d-bit barcode ids do not fit the packed sort key (events < 2^%d): load the reads in windows", (long long)n_events, bits, 52 - bits + src_shift);
        return -2;
    }
    // A load that is counted once, without a store, by a count that admits every stored read (the filters of the load) and whose depth cap
    // cannot fire needs nothing of an entry's value but two flags: they ride in the key (bits 62, 63, above a source field two bits
    // narrower) and the scatter, the sort and the count move 8 bytes an entry instead of 12.  Should the count not be made that way after
    // all (its rows outgrow their buffer), the load starts again with values (keys_only_off).
    bool keys_only = c->cal_enabled && c->store_policy == LSG_STORE_SKIP_WHEN_COUNTED && !getenv("LSG_NO_DIRECT_COUNT") && !getenv("LSG_NO_FUSED_LOAD") &&
                     !getenv("LSG_NO_KEYS_ONLY") && !c->keys_only_off && c->n_ct >= 1 && c->n_ct <= 2 && c->n_cb > 0 && c->copy_stream && n_events >= 64 &&
                     n_events < (1ll << 39) && (n_events >> src_shift) < (1ll << (50 - bits - 2 * wsh));
    // Can the depth cap of the count this load is to make fire at all?  The all-reads bound per tile (window) came with the early look; when it
    // cannot say no, the per-position bound of this table's cell types decides (layout.hip: ~10 ms at C4, whose tiles hold more than 200 000
    // reads that no position does) - here, before the scatter, because a count whose cap cannot fire may sort keys alone.
    bool cap_out = true;
    if (c->cal_enabled && c->cal_params.max_depth > 0 && c->max_live_all + 1 > (int64_t)c->cal_params.max_depth) {
        cap_out = false;
        if (c->n_ct >= 1 && c->n_ct <= 2 && c->n_cb > 0 && c->copy_stream && !getenv("LSG_NO_FUSED_LOAD")) {
            LSG_HIP(hipStreamSynchronize(c->copy_stream));          // (the handle's copies of the read arrays, which the bounds read, are made there)
            if (live_read_bound(c)) return -1;                      // first the tiles again, per cell type of this table (C4: 135 343 against 225 294 over all reads)
            cap_out = c->max_live_reads + 1 <= (int64_t)c->cal_params.max_depth;
            if (!cap_out) {
                if (live_read_bound_exact(c)) return -1;
                cap_out = c->max_live_exact <= (int64_t)c->cal_params.max_depth;
            }
        }
    }
    if (keys_only) {
        const lsg_count_params& q = c->cal_params;
        if (q.min_mq != c->st_min_mq || q.flag_exclude != c->st_flag_exclude || (q.ignore_orphans != 0) != (c->st_ignore_orphans != 0)) keys_only = false;
        if (!cap_out) keys_only = false;
        for (int t = 0; t < c->n_contigs && keys_only; ++t) if (!c->ref_ptr[t]) keys_only = false;
    }
    if (win && !keys_only) return start_again_by_tiles();
    if (keys_only && getenv("LSG_TIMING")) fprintf(stderr, "[lsg] load: keys alone through the scatter and the sort (8 bytes an entry)%s\n", win ? ", entries binned by 128-position windows" : "");
    if (key_a.reserve(N * 8 + 16) || key_b.reserve(N * 8 + 16) || (!keys_only && (val_a.reserve(N * 4 + 16) || val_b.reserve(N * 4 + 16))) ||
        c->bt[BT_CURSOR].reserve(((size_t)T + 2) * 4)) return -1;
    // (the cursors in a buffer of their own: the plan's tile-level half may be at work in BT_PER_TILE beside the scatter)
    a.cursor = c->bt[BT_CURSOR].as<uint32_t>(); a.key = key_a.as<uint64_t>(); a.rdv = keys_only ? nullptr : val_a.as<uint32_t>(); a.cb_bits = bits; a.src_shift = src_shift;
    LSG_HIP(hipMemcpyAsync(a.cursor, c->d_tile_off.p, ((size_t)T + 1) * 4, hipMemcpyDeviceToDevice, st));
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, 8, st));
    hipLaunchKernelGGL(k_bin, dim3(g_bin), dim3(BIN_THREADS), 0, st, a);
    LSG_HIP(hipEventRecord(c->evb[1], st));
    // ---- beside the scatter, on the copy stream: what follows from the capacities alone
    if (c->tm[TM_BLK_TILE].reserve(((size_t)nblk + 2) * 4)) return -1;
    bool lpt = false, split_sort = false, blk_tiles_made = false;
    const uint32_t* lpt_tiles = nullptr;
    const bool may_skip_store = c->cal_enabled && c->store_policy == LSG_STORE_SKIP_WHEN_COUNTED && !getenv("LSG_NO_DIRECT_COUNT");
    // (the scratch is BT_COPY_TMP, sized below before anything is queued on the copy stream; on the main stream - the fall-back - the
    // copy stream has been waited for)
    auto blk_tiles = [&](hipStream_t s_) -> int {
        uint32_t* bt_ = c->tm[TM_BLK_TILE].as<uint32_t>();
        LSG_HIP(hipMemsetAsync(bt_, 0, (size_t)nblk * 4, s_));
        hipLaunchKernelGGL(k_tm_blk_mark, dim3((T + 255) / 256), dim3(256), 0, s_, c->d_tile_cap.as<uint32_t>(), blk_off, T, bt_);
        size_t tb = c->bt[BT_COPY_TMP].cap;
        LSG_HIP(hipcub::DeviceScan::InclusiveScan(c->bt[BT_COPY_TMP].p, tb, bt_, bt_, hipcub::Max(), (int)nblk, s_));
        LSG_HIP(hipEventRecord(c->ev_blk, s_));
        blk_tiles_made = true;
        return 0;
    };
    {
        hipStream_t bs = c->copy_stream;
        const bool want_lpt = n_netile > 1 && !getenv("LSG_NO_LPT");
        DevBuf& lk = c->bt[BT_LPT];
        const size_t lpt_pitch = ((size_t)n_netile + 64) & ~(size_t)63;            // (every array on a 256-byte boundary)
        if (want_lpt && lk.reserve(lpt_pitch * 4 * 3)) return -1;
        uint32_t* k_in = lk.as<uint32_t>(); uint32_t* k_out = k_in + lpt_pitch; uint32_t* t_out = k_out + lpt_pitch;
        uint32_t* bt_ = c->tm[TM_BLK_TILE].as<uint32_t>();
        // (the scratch of both is sized BEFORE either is queued: growing a buffer frees it, and work queued on this stream may still be reading it)
        size_t tb_lpt = 0, tb_blk = 0;
        if (want_lpt) LSG_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb_lpt, k_in, k_out, c->bt[BT_NETILE].as<uint32_t>(), t_out, (int)n_netile, 0, 21, bs));
        LSG_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, tb_blk, bt_, bt_, hipcub::Max(), (int)nblk, bs));
        // (... and of the sort of the shallow tiles, which runs on this stream beside the deep tiles' sort: below)
        size_t tb_sort2 = 0;
        if (n_netile && !getenv("LSG_NO_SPLIT_SORT")) {
            if (keys_only) LSG_HIP((lsg_segmented_sort_keys<7, 256, 8>(nullptr, tb_sort2, (uint64_t*)nullptr, (uint64_t*)nullptr, (unsigned)N, (unsigned)n_netile, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned)bits, bs, false)));
            else LSG_HIP((lsg_segmented_sort<7, 256, 8>(nullptr, tb_sort2, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned)N, (unsigned)n_netile,
                                                        (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (unsigned)bits, bs, false)));
        }
        split_sort = tb_sort2 != 0;
        size_t tb_max = tb_lpt > tb_blk ? tb_lpt : tb_blk;
        if (tb_sort2 > tb_max) tb_max = tb_sort2;
        if (c->bt[BT_COPY_TMP].reserve(tb_max + 256)) return -1;
        void* scratch = c->bt[BT_COPY_TMP].p;
        if (want_lpt) {
            // The sort's segments deepest first: rocprim gives every segment beyond its block sort to ONE workgroup, and the deepest tiles
            // (chrM: the last contig) would start last - their workgroups' tail is then hidden behind the rest.  A stable radix sort of the
            // non-empty tiles by capacity >> 11 (what lies above the block sort's 2048: the shallower ones keep their order).  The keys are
            // shifted by k_tile_caps and sorted from bit 0: rocprim 7.2's radix sort with begin_bit > 0 returns wrong pairs below ~1M items.
            hipLaunchKernelGGL(k_tile_caps, dim3((n_netile + 255) / 256), dim3(256), 0, bs, c->bt[BT_NETILE].as<uint32_t>(), n_netile, c->d_tile_cap.as<uint32_t>(), k_in);
            size_t tb = c->bt[BT_COPY_TMP].cap;
            LSG_HIP(hipcub::DeviceRadixSort::SortPairsDescending(scratch, tb, k_in, k_out, c->bt[BT_NETILE].as<uint32_t>(), t_out, (int)n_netile, 0, 21, bs));
            LSG_HIP(hipEventRecord(c->ev_lpt, bs));
            lpt = true; lpt_tiles = t_out;
        }
        // the tile of every block (the re-counts' resolve and the plain gather read it; the fused pass does not) - left out when the load is
        // to keep no store (should it build one after all, blk_tiles() runs then)
        if (!may_skip_store) { if (int rc = blk_tiles(bs)) return rc; }
    }
    // The plan's tile-level half (units, jobs and slabs per bin and their totals: it reads the capacities only) on the copy stream NOW, beside
    // the scatter - it ends in a host round trip that used to wait for the shallow bins' sort on that stream (0.25 ms of nothing between the
    // sort and the count for a rank's eighth of a sample) - whenever the load is to make its count
    bool tiles_planned = false;
    if (c->cal_enabled && c->n_ct >= 1 && c->n_ct <= 2 && c->n_cb > 0 && c->copy_stream && !getenv("LSG_NO_FUSED_LOAD") && !getenv("LSG_LATE_PLAN")) {
        if (int rc = plan_tiles(c, c->copy_stream)) return rc;
        plan_finish_tiles(c);
        tiles_planned = true;
    }
    // ---- 3. every tile's entries by barcode
    if (n_netile) {
        // (begin / end of the deep tiles' sort, then - split_sort - of the shallow tiles' sort: four arrays of n_netile + 1 in the two buffers)
        const size_t seg_pitch = ((size_t)n_netile + 64) & ~(size_t)63;
        if (c->bt[BT_SEG_BEGIN].reserve(seg_pitch * 8) || c->bt[BT_SEG_END].reserve(seg_pitch * 8)) return -1;
        uint32_t* sb1 = c->bt[BT_SEG_BEGIN].as<uint32_t>(); uint32_t* se1 = c->bt[BT_SEG_END].as<uint32_t>();
        uint32_t* sb2 = split_sort ? sb1 + seg_pitch : nullptr; uint32_t* se2 = split_sort ? se1 + seg_pitch : nullptr;
        if (lpt) LSG_HIP(hipStreamWaitEvent(st, c->ev_lpt, 0));
        hipLaunchKernelGGL(k_seg_bounds, dim3((n_netile + 255) / 256), dim3(256), 0, st, lpt ? lpt_tiles : c->bt[BT_NETILE].as<uint32_t>(), n_netile, c->d_tile_off.as<uint32_t>(), sb1, se1, sb2, se2);
        if (split_sort) LSG_HIP(hipEventRecord(c->ev_copy, st));      // (the second sort, queued below, starts here)
        size_t tb = 0;
        // (64-bit keys sorted on their barcode bits only, begin_bit 0 .. end_bit `bits`: the rest of the key is payload)
        // (one configuration: 7 bits per pass, 256 x 8 items in the block sort - the sweep over 512 / 1024 threads, 4-16 items and 8 bits per pass
        // that used to be selectable here found nothing faster, and cost a minute of compile time)
        const char* bb_env = getenv("LSG_SORT_BIG_BLOCKS");
        const bool big_blocks = bb_env ? atoi(bb_env) != 0 : N < (64ull << 20);       // (C2: 23 M entries a rank in eight shards, 46 M in four, 92 M in two - where the big workgroups already lose)
        auto sort = [&](void* tmp_p, size_t& tmp_n) {
            // (a small load - a rank's share of a sharded sample - is bound by ONE workgroup working through its deepest tile, 2048 entries
            // an iteration: 512 threads take 4096 - C2 in eight shards 0.71 -> 0.42-0.55 ms, in four 0.76 -> 0.69; 1024 threads gain on the
            // deepest tiles and lose on all others; a whole sample is bound by throughput, where the small workgroups win)
            if (keys_only && big_blocks) return lsg_segmented_sort_keys<7, 512, 8>(tmp_p, tmp_n, key_a.as<uint64_t>(), key_b.as<uint64_t>(), (unsigned)N, (unsigned)n_netile, sb1, se1, 0u, (unsigned)bits, st, false);
            if (keys_only) return lsg_segmented_sort_keys<7, 256, 8>(tmp_p, tmp_n, key_a.as<uint64_t>(), key_b.as<uint64_t>(), (unsigned)N, (unsigned)n_netile, sb1, se1, 0u, (unsigned)bits, st, false);
            return lsg_segmented_sort<7, 256, 8>(tmp_p, tmp_n, key_a.as<uint64_t>(), key_b.as<uint64_t>(), val_a.as<uint32_t>(), val_b.as<uint32_t>(), (unsigned)N, (unsigned)n_netile,
                                                 sb1, se1, 0u, (unsigned)bits, st, false);
        };
        LSG_HIP(sort(nullptr, tb));
        if (tmp.reserve(tb + 256)) return -1;
        tb = tmp.cap;
        LSG_HIP(sort(tmp.p, tb));
        if (split_sort) {
            // rocprim runs its three kernels (a block per segment beyond 256 entries: 2.7 ms at C2; the warp sorts of the segments up to 256 and up
            // to 64: 0.55 ms) one after the other, and the first one's tail leaves most of the chip idle: the shallow tiles are sorted on the copy
            // stream beside it (same output arrays, disjoint segments, scratch of its own); queued AFTER the deep tiles' sort so that the
            // main stream does not wait for these launches
            hipStream_t bs = c->copy_stream;
            LSG_HIP(hipStreamWaitEvent(bs, c->ev_copy, 0));
            size_t tb2 = c->bt[BT_COPY_TMP].cap;
            if (keys_only) LSG_HIP((lsg_segmented_sort_keys<7, 256, 8>(c->bt[BT_COPY_TMP].p, tb2, key_a.as<uint64_t>(), key_b.as<uint64_t>(), (unsigned)N, (unsigned)n_netile, sb2, se2, 0u, (unsigned)bits, bs, false)));
            else LSG_HIP((lsg_segmented_sort<7, 256, 8>(c->bt[BT_COPY_TMP].p, tb2, key_a.as<uint64_t>(), key_b.as<uint64_t>(), val_a.as<uint32_t>(), val_b.as<uint32_t>(), (unsigned)N, (unsigned)n_netile,
                                                        sb2, se2, 0u, (unsigned)bits, bs, false)));
            LSG_HIP(hipEventRecord(c->ev_lpt, bs));            // (the event of the tiles' order: waited for above, free again)
        }
        if (split_sort) LSG_HIP(hipStreamWaitEvent(st, c->ev_lpt, 0));      // both halves of the order are there
    }
    LSG_HIP(hipEventRecord(c->evb[2], st));
    // ---- 4. blocks and the per-entry words (the blocks' offsets and their number came with the load's early look)
    const uint64_t np = (uint64_t)nblk * 8;
    c->tm_np = np; c->tm_nblk = nblk;
    auto reserve_store = [&]() -> int {
    if (c->tm[TM_S0].reserve((np + 16) * 4) || c->tm[TM_B].reserve(np + 16) ||
        c->tm[TM_RD].reserve((np + 16) * 4) || c->tm[TM_META].reserve((np + 8 * (TM_GROUP + 1)) * 4) ||
        c->tm[TM_BLK_TILE].reserve(((size_t)nblk + 2) * 4) || c->tm[TM_EXT].reserve(((size_t)nblk + TM_GROUP + 2) * 2)) return -1;
    LSG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->tm[TM_S0].as<uint32_t>() + np), (int)TM_PAD_S0, 16, st));       // (what the walk's group loads and the run flags' neighbours see past the end)
    LSG_HIP(hipMemsetAsync(c->tm[TM_B].as<uint8_t>() + np, 0, 16, st));
    // ---- 5. the per-entry words and the events.  The blocks are the load's largest allocation (C4: 124 GB beside 106 GB of the caller's
    // events): they are reserved once and kept; when the device is nearly full what the build no longer needs goes first
    if (c->tm[TM_STORE].cap < ((size_t)nblk + TM_GROUP) * 1024) {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && ((size_t)nblk + TM_GROUP) * 1088 + (mem_total >> 5) > mem_free) {
            LSG_HIP(hipStreamSynchronize(st));
            key_a.release(); val_a.release(); c->ws[WS_SEG_INFO].release(); c->bt[BT_SPAN].release(); c->bt[BT_SPAN_RUN].release(); c->bt[BT_PEX].release(); c->bt[BT_OFFS].release();
        }
        if (c->tm[TM_STORE].reserve(((size_t)nblk + TM_GROUP) * 1024)) return -1;
    }
    return 0;
    };
    LSG_HIP(hipEventRecord(c->evb[3], st));
    // The first count in the same pass (lsg_set_count_at_load): when its parameters and the barcode table are known now, the depth cap
    // cannot fire (the all-reads bound came with the early look) and one pass covers the cell types, the gather below is replaced by
    // pileup.hip's k_tm_gather_count, which builds the same store and counts while each block is in registers.
    bool fused = c->cal_enabled && c->n_ct >= 1 && c->n_ct <= 2 && c->n_cb > 0 && c->copy_stream && n_events >= 64 && n_events < (1ll << 39) && !getenv("LSG_NO_FUSED_LOAD");
    if (fused) {
        const lsg_count_params& q = c->cal_params;
        if (q.min_mq < c->st_min_mq || (c->st_flag_exclude & ~q.flag_exclude) != 0 || (c->st_ignore_orphans && !q.ignore_orphans)) fused = false;       // (the count would be refused)
        if (fused && !cap_out) fused = false;                     // the depth cap might fire (decided before the scatter, above): the count is left to lsg_pileup_count
        for (int t = 0; t < c->n_contigs && fused; ++t) if (!c->ref_ptr[t]) fused = false;
    }
    const bool dbg = getenv("LSG_DEBUG_SYNC") != nullptr;
    auto stage = [&](const char* what) { if (dbg) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[lsg] fused load: %s: %s\n", what, hipGetErrorString(e)); fflush(stderr); } };
    // the plan: its tile-level half on the copy stream while the scatter and the sort are at work (two small host round trips that
    // wait for the copy stream only), its job-level half behind the sort - the jobs are cut at run starts read from the sorted keys
    auto fused_plan = [&]() -> int {
        stage("scatter + sort + block tables");
        int rc = 0;
        if (!tiles_planned) {
            if ((rc = plan_tiles(c, c->copy_stream))) return rc;
            plan_finish_tiles(c);
        }
        c->tm_np = np; c->tm_nblk = nblk;
        if ((rc = plan_jobs(c, false, key_b.as<uint64_t>(), bits))) return rc;
        stage("plan");
        return 0;
    };
    const GatherCountSrc gsrc{events, n_events, key_b.as<uint64_t>(), keys_only ? nullptr : val_b.as<uint32_t>(), bits, src_shift, wsh};
    auto load_again_with_values = [&]() -> int {                 // (a load of keys alone that is not counted that way after all)
        LSG_HIP(hipStreamSynchronize(st)); LSG_HIP(hipStreamSynchronize(c->copy_stream));
        if (getenv("LSG_TIMING")) fprintf(stderr, "[lsg] load: the count from keys alone was not made, loading again with values\n");
        c->keys_only_off = true;
        const int rc = build_store(c, events, n_events, seg_ev_off, src);
        c->keys_only_off = false;
        return rc;
    };
    bool planned = false;
    if (win && !fused) return start_again_by_tiles();
    if (keys_only && getenv("LSG_TEST_KEYS_ONLY_REFUSED")) return win ? start_again_by_tiles() : load_again_with_values();      // (test hook: the way a refused count of keys alone takes)
    if (fused && c->store_policy == LSG_STORE_SKIP_WHEN_COUNTED && !getenv("LSG_NO_DIRECT_COUNT")) {
        // A load that is counted once and never again (lsg_set_store_policy): the count alone, from the caller's events through the
        // sort's output - no blocks, no per-entry words.  What needs a store afterwards is refused until the next load.
        if (int rc = fused_plan()) return rc;
        planned = true;
        const int rc = run_gather_count(c, &c->cal_params, gsrc, true);
        LSG_HIP(hipStreamSynchronize(st));
        if (hipGetLastError() != hipSuccess) { set_error("lsg_load_reads: the count pass failed"); return -1; }
        if (rc < 0 && rc != -3) return rc;                        // a failure (allocation, HIP) is one: only rows that outgrew their buffer (-3) or a refused count of keys alone (1) load again
        if (rc == 0) {
            plan_finish(c);
            LSG_HIP(hipStreamSynchronize(c->copy_stream));     // (the blocks' tiles: nobody reads them, nothing may still be writing them)
            c->load_was_fused = true; c->counted_at_load = true; c->store_skipped = true;
            for (int i = 0; i < 4; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, c->evb[i], c->evb[i + 1]) == hipSuccess) c->build_ms[i] = ms; }
            settle_temporaries(c);
            c->layout_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
            return 0;                                             // (tm_valid stays false: there is no store)
        }
        c->counted = false;                                       // rows outgrew their buffer: the store is built after all, the count is made on request
        fused = false;
    }
    if (win) return start_again_by_tiles();                  // (the windows' count was not made: rows outgrew their buffer, a count that is not the load's ...)
    if (keys_only) return load_again_with_values();
    if (int rc = reserve_store()) return rc;
    if (!blk_tiles_made) {                                       // (a load that was to keep no store builds one after all)
        LSG_HIP(hipStreamSynchronize(c->copy_stream));
        if (int rc = blk_tiles(st)) return rc;
    }
    if (fused) {
        if (int rc = fused_plan()) return rc;
        c->tm_valid = true;                                   // (what the count's preparation looks at; the blocks are written by the pass itself)
        int rc = run_gather_count(c, &c->cal_params, gsrc, false);
        LSG_HIP(hipStreamSynchronize(st));
        if (hipGetLastError() != hipSuccess) { c->tm_valid = false; set_error("lsg_load_reads: the fused gather + count pass failed"); return -1; }
        if (rc < 0 && rc != -3) { c->tm_valid = false; return rc; }      // (as above: -3 = the rows outgrew their buffer, the store is whole and counted on request)
        plan_finish(c);
        c->load_was_fused = true;
        c->counted_at_load = rc == 0;                         // (a count that could not be kept - rows outgrew their buffer - is made again on request; the store is whole either way)
        if (rc) c->counted = false;
    } else {
    LSG_HIP(hipStreamWaitEvent(st, c->ev_blk, 0));                  // (the blocks' tiles, made on the copy stream)
    const bool plan_early = c->n_ct > 0 && c->copy_stream && c->ev_copy && !planned;      // (planned: a count without a store was tried and its rows did not fit)
    if (plan_early) LSG_HIP(hipEventRecord(c->ev_copy, st));       // everything the gather waits for is what the plan's tile-level half waits for
    {
        const dim3 grid((unsigned)(((((uint64_t)nblk + TMG_BLOCKS - 1) / TMG_BLOCKS + TMG_WAVES - 1) / TMG_WAVES + 7) / 8 * 8));
        hipLaunchKernelGGL(k_tm_gather, grid, dim3(TMG_WAVES * 64), 0, st,
                           events, n_events, key_b.as<uint64_t>(), val_b.as<uint32_t>(), bits, src_shift, c->d_tile_off.as<uint32_t>(), blk_off, c->tm[TM_BLK_TILE].as<uint32_t>(), nblk,
                           c->tm[TM_S0].as<uint32_t>(), c->tm[TM_B].as<uint8_t>(), c->tm[TM_RD].as<uint32_t>(),
                           c->tm[TM_STORE].as<uint4>(), c->tm[TM_EXT].as<uint16_t>());
    }
    LSG_HIP(hipEventRecord(c->evb[4], st));
    LSG_HIP(hipGetLastError());
    int plan_rc = 0;
    if (plan_early) {       // beside the gather, on the copy stream: a dozen small kernels and two host round trips that the first count would otherwise pay
        LSG_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
        plan_rc = tiles_planned ? 0 : plan_tiles(c, c->copy_stream);
        if (!plan_rc) {        // ... and its job-level half right behind the gather: the load's last synchronisation is the plan's too
            c->tm_np = np; c->tm_nblk = nblk;
            plan_rc = plan_jobs(c, false);
        }
    }
    LSG_HIP(hipStreamSynchronize(st));
    if (plan_early) LSG_HIP(hipStreamSynchronize(c->copy_stream));
    if (plan_rc) return plan_rc;
    if (plan_early || planned) plan_finish(c);
    }
    for (int i = 0; i < 4; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, c->evb[i], c->evb[i + 1]) == hipSuccess) c->build_ms[i] = ms; }
    settle_temporaries(c);
    if (getenv("LSG_TIMING"))
        fprintf(stderr, "[lsg] tile store: %llu entries, %u blocks (%.2f GB): capacities + scatter %.2f, sort %.2f, fill %.2f, gather %.2f ms\n",
                (unsigned long long)N, nblk, (double)nblk * 1024 / 1e9, c->build_ms[0], c->build_ms[1], c->build_ms[2], c->build_ms[3]);
    return finish();
}

// ================================================================================================
// The plan of a count over the store: everything about jobs, units and slabs is static per (load, number of cell types), so a count
// has no planning step and ONE host synchronisation (its final read of the counters).  Unit = (tile, cell type), both cell types of a
// pass counted by one job (a barcode's run belongs to one cell type).  Tiles of more than TM_JOB_TGT entries are cut at run starts into
// jobs of about TM_JOB_TGT entries (partial sums to slabs that k_finalize_multi adds: 8 KB per job and cell type, so jobs are as long
// as the planes' fields allow); a job that a single barcode's run stretches past TM_JOB_LIMIT goes to the wide walk.
constexpr uint32_t TM_CHUNK_WORK = 4096, TM_JOB_W0 = 32;      // a workgroup dequeues at most this much work (entries + a constant per job) at a time

// per tile: non-empty, jobs, slabs, multi-job (inputs of four exclusive scans)
// (H: tiles per bin - 2 when the bins are 128-position windows: a non-empty window has the units of BOTH its tiles)
__global__ void k_tm_tiles(const uint32_t* cap, uint32_t n_tiles, int n_ct, uint32_t job_tgt, uint32_t H, uint32_t* ne, uint32_t* nj, uint32_t* slabs, uint32_t* multi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    const uint32_t n = t < n_tiles ? cap[t] : 0u;
    const uint32_t j = n == 0 ? 0u : (n <= job_tgt ? 1u : (n + job_tgt - 1) / job_tgt);
    ne[t] = n ? H : 0u; nj[t] = j; slabs[t] = j > 1 ? j * (uint32_t)n_ct * H : 0u; multi[t] = j > 1 ? H : 0u;
}
// job j of J of a tile of n entries at `base`: [first run start at or after n j / J, first run start at or after n (j + 1) / J)
// (run starts: the store's s0 words, or - before the gather has written them - the sorted keys' barcodes; RunSrc)
struct RunSrc { const uint32_t* s0; const uint64_t* key; uint32_t cbm; };
__device__ __forceinline__ void tm_make_job(const RunSrc& rs, uint64_t base, uint32_t off, uint32_t n, uint32_t J, uint32_t j, uint32_t w0, uint32_t slab, uint32_t tile, int2 geom,
                                            TmJob* out, uint32_t* n_wide) {
    auto starts = [&](uint32_t x) -> bool {
        if (!rs.key) return (rs.s0[base + x] & TM_RUNSTART) != 0;
        return x == 0 || (((uint32_t)rs.key[off + x] ^ (uint32_t)rs.key[off + x - 1]) & rs.cbm) != 0;
    };
    auto cut = [&](uint32_t x) -> uint32_t { while (x < n && !starts(x)) ++x; return x < n ? x : n; };
    const uint32_t e0 = j == 0 ? 0u : cut((uint32_t)(((uint64_t)n * j) / J));
    uint32_t e1 = j + 1 == J ? n : cut((uint32_t)(((uint64_t)n * (j + 1)) / J));
    if (e1 < e0) e1 = e0;
    TmJob jb;
    jb.e0 = (uint32_t)(base + e0); jb.e1 = (uint32_t)(base + e1); jb.w0 = w0;
    jb.slab = J > 1 ? slab : 0xFFFFFFFFu; jb.nj = J; jb.cnt = n; jb.tile = tile;
    jb.base = (uint32_t)base; jb.off = off; jb.tstart = geom.x; jb.tid = geom.y & 0xffffff;
    // two waves share the job: the second starts at the run start at or after its middle (short jobs: one wave)
    uint32_t mid = e1;
    if (e1 - e0 >= 64u) { mid = cut(e0 + (e1 - e0) / 2u); if (mid > e1) mid = e1; }
    jb.emid = (uint32_t)(base + mid);
    if (e1 - e0 > (uint32_t)TM_JOB_LIMIT) { jb.nj |= TMJ_WIDE; atomicAdd(n_wide, 1u); }
    *out = jb;
}
// per non-empty tile: its units (one per cell type) and, for a tile that is one job, the job
// (wsh = 1: t is a window; its units are those of tiles 2 t and 2 t + 1, tile by tile)
__global__ void k_tm_jobs(const uint32_t* tile_base, int n_contigs, int n_ct, int wsh, RunSrc rs, const uint32_t* tile_off, const uint32_t* cap, const uint32_t* blk_off, const uint32_t* ne_off,
                          const uint32_t* nj, const uint32_t* job_off, const uint32_t* slab_off, const uint32_t* multi_off, uint32_t n_tiles, TmJob* jobs,
                          uint32_t* ne_units, int2* ne_geom, uint32_t* ne_nslot, uint32_t* ne_acc, uint32_t* multi, uint32_t* n_wide) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t n = cap[t];
    if (!n) return;
    const uint32_t J = nj[t], ord = ne_off[t];
    const uint64_t base = (uint64_t)blk_off[t] * 8;
    const uint32_t tile0 = t << wsh;
    int tid = 0;
    { int lo = 0, hi = n_contigs; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_base[mid] <= tile0) lo = mid; else hi = mid; } tid = lo; }
    const int32_t tstart = (int32_t)((tile0 - tile_base[tid]) << 6);
    for (uint32_t h = 0; h < (1u << wsh); ++h)
    for (int ct = 0; ct < n_ct; ++ct) {
        const uint32_t w = (ord + h) * (uint32_t)n_ct + ct;
        ne_units[w] = (tile0 + h) * (uint32_t)n_ct + ct;
        ne_geom[w] = make_int2(tstart + 64 * (int32_t)h, tid | (ct << 24));
        ne_nslot[w] = J;
        ne_acc[w] = J > 1 ? slab_off[t] + (h * (uint32_t)n_ct + (uint32_t)ct) * J : 0u;
        if (J > 1) multi[(multi_off[t] + h) * (uint32_t)n_ct + ct] = w;
    }
    if (J > 1) return;                                   // its jobs: k_tm_jobs_multi (a lane per job; here one thread would walk them one after the other)
    tm_make_job(rs, base, tile_off[t], n, 1u, 0u, ord * (uint32_t)n_ct, 0xFFFFFFFFu, t, make_int2(tstart, tid), jobs + job_off[t], n_wide);
}
// the jobs of the tiles cut into several: a wave per tile, a lane per job (a job's ends are run starts found from its own nominal ends:
// no job waits for the one before it)
__global__ __launch_bounds__(64) void k_tm_jobs_multi(int n_ct, int wsh, RunSrc rs, const uint32_t* tile_off, const uint32_t* cap, const uint32_t* blk_off, const uint32_t* ne_off, const uint32_t* nj,
                                                      const uint32_t* job_off, const uint32_t* slab_off, const uint32_t* multi, const uint32_t* ne_units, const int2* ne_geom, uint32_t n_mt,
                                                      TmJob* jobs, uint32_t* n_wide) {
    const uint32_t i = blockIdx.x;
    if (i >= n_mt) return;
    const uint32_t w = multi[((size_t)i << wsh) * (uint32_t)n_ct], t = (ne_units[w] / (uint32_t)n_ct) >> wsh;      // (the bin's first unit)
    const uint32_t n = cap[t], J = nj[t];
    const int2 geom = ne_geom[w];                        // (k_tm_jobs wrote the tile's units before this kernel started)
    const uint64_t base = (uint64_t)blk_off[t] * 8;
    for (uint32_t j = threadIdx.x; j < J; j += 64u)
        tm_make_job(rs, base, tile_off[t], n, J, j, ne_off[t] * (uint32_t)n_ct, slab_off[t] + j, t, geom, jobs + job_off[t] + j, n_wide);
}
struct TmJobWork {
    const TmJob* jobs;
    __host__ __device__ uint32_t operator()(const uint32_t& j) const { return jobs[j].e1 - jobs[j].e0 + TM_JOB_W0; }
};
// chunk k = the jobs whose exclusive work prefix lies in [k E, (k + 1) E)
__global__ void k_tm_chunks(const uint32_t* pex, uint32_t njobs, uint32_t chunk_work, uint32_t* chunk_start, uint32_t* n_chunks) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    const uint32_t ck = pex[j] / chunk_work;
    const int64_t prev = j ? (int64_t)(pex[j - 1] / chunk_work) : -1;
    for (int64_t k = prev + 1; k <= (int64_t)ck; ++k) chunk_start[k] = j;
    if (j == njobs - 1) { chunk_start[ck + 1] = njobs; *n_chunks = ck + 1; }
}

// The plan's tile-level half: units, jobs, slabs and multi-job marks per tile, their running sums and totals, the buffers those totals
// size.  It reads the tiles' capacities only, so the load makes it on the copy stream BESIDE its gather (which then is all the device
// is waiting for) when the number of cell types is already known; else the first count makes it.
static int plan_tiles(lsg_ctx* c, hipStream_t st) {
    const uint32_t T = c->n_tiles >> c->wsh;      // (bins: tiles, or windows of two)
    const uint64_t N = c->tm_n;
    DevBuf &per_tile = c->bt[BT_PER_TILE], &offs = c->bt[BT_OFFS];
    if (per_tile.reserve((size_t)(T + 2) * 4 * 4) || offs.reserve((size_t)(T + 2) * 4 * 4 + 64)) return -1;
    uint32_t* ne = per_tile.as<uint32_t>(); uint32_t* nj = ne + (T + 2); uint32_t* slabs = nj + (T + 2); uint32_t* multi = slabs + (T + 2);
    uint32_t* ne_off = offs.as<uint32_t>(); uint32_t* job_off = ne_off + (T + 2); uint32_t* slab_off = job_off + (T + 2); uint32_t* multi_off = slab_off + (T + 2);
    uint32_t* d_misc = multi_off + (T + 2);          // [0] wide jobs, [1] chunks
    // jobs as long as the planes' fields allow (fewer slabs) — unless the load is small (one rank's share of a sharded job): then every
    // resident pair of waves should still get several
    uint32_t job_tgt = TM_JOB_TGT;
    { const uint64_t per = N / ((uint64_t)c->n_cus * 14 * 4); if (per < job_tgt) job_tgt = (uint32_t)(per < 768 ? 768 : per); }
    hipLaunchKernelGGL(k_tm_tiles, dim3((T + 256) / 256), dim3(256), 0, st, c->d_tile_cap.as<uint32_t>(), T, c->n_ct, job_tgt, 1u << c->wsh, ne, nj, slabs, multi);
    SCAN_U32(ne, ne_off, T + 1); SCAN_U32(nj, job_off, T + 1); SCAN_U32(slabs, slab_off, T + 1); SCAN_U32(multi, multi_off, T + 1);
    LSG_HIP(hipMemsetAsync(d_misc, 0, 8, st));
    uint32_t* srcs[4] = {ne_off, job_off, slab_off, multi_off};
    for (int i = 0; i < 4; ++i) LSG_HIP(hipMemcpyAsync(&c->plan1_tot[i], srcs[i] + T, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    const uint32_t n_net = c->plan1_tot[0], njobs = c->plan1_tot[1], n_mt = c->plan1_tot[3];
    const size_t n_ne = (size_t)n_net * (size_t)c->n_ct, n_multi = (size_t)n_mt * (size_t)c->n_ct;
    if (c->tm[TM_JOBS].reserve(((size_t)njobs + 1) * sizeof(TmJob)) || c->tm[TM_NE_UNITS].reserve((n_ne + 2) * 4) || c->tm[TM_NE_GEOM].reserve((n_ne + 2) * 8) ||
        c->tm[TM_NE_NSLOT].reserve((n_ne + 2) * 4) || c->tm[TM_NE_ACC].reserve((n_ne + 2) * 4) || c->tm[TM_MULTI].reserve((n_multi + 2) * 4)) return -1;
    LSG_HIP(hipMemsetAsync(c->tm[TM_NE_NSLOT].p, 0, (n_ne + 2) * 4, st));
    LSG_HIP(hipMemsetAsync(c->tm[TM_NE_ACC].p, 0, (n_ne + 2) * 4, st));
    LSG_HIP(hipStreamSynchronize(st));
    c->plan1_n_ct = c->n_ct;
    return 0;
}

// The plan's job-level half: jobs cut at the run starts the gather wrote, the units' tables, the walk's work-balanced chunks.  Queued
// behind the gather by the load when the tile-level half is there (finish = false: the load's own final synchronisation covers it and
// plan_finish reads the two counters), else made by the first count.
static int plan_jobs(lsg_ctx* c, bool finish, const uint64_t* skey, int cb_bits) {
    hipStream_t st = c->stream;
    const RunSrc rs{c->tm[TM_S0].as<uint32_t>(), skey, skey ? (1u << cb_bits) - 1u : 0u};
    const uint32_t T = c->n_tiles >> c->wsh;
    DevBuf &per_tile = c->bt[BT_PER_TILE], &offs = c->bt[BT_OFFS];
    uint32_t* nj = per_tile.as<uint32_t>() + (T + 2);
    uint32_t* ne_off = offs.as<uint32_t>(); uint32_t* job_off = ne_off + (T + 2); uint32_t* slab_off = job_off + (T + 2); uint32_t* multi_off = slab_off + (T + 2);
    uint32_t* d_misc = multi_off + (T + 2);          // [0] wide jobs, [1] chunks
    const uint32_t* tot = c->plan1_tot;
    const uint32_t njobs = tot[1], n_mt = tot[3];
    hipLaunchKernelGGL(k_tm_jobs, dim3((T + 255) / 256), dim3(256), 0, st, c->d_tile_base.as<uint32_t>(), c->n_contigs, c->n_ct, c->wsh, rs, c->d_tile_off.as<uint32_t>(),
                       c->d_tile_cap.as<uint32_t>(), c->tm[TM_BLK_OFF].as<uint32_t>(), ne_off, nj, job_off, slab_off, multi_off, T,
                       c->tm[TM_JOBS].as<TmJob>(), c->tm[TM_NE_UNITS].as<uint32_t>(), c->tm[TM_NE_GEOM].as<int2>(), c->tm[TM_NE_NSLOT].as<uint32_t>(),
                       c->tm[TM_NE_ACC].as<uint32_t>(), c->tm[TM_MULTI].as<uint32_t>(), d_misc);
    if (n_mt) hipLaunchKernelGGL(k_tm_jobs_multi, dim3(n_mt >> c->wsh), dim3(64), 0, st, c->n_ct, c->wsh, rs, c->d_tile_off.as<uint32_t>(), c->d_tile_cap.as<uint32_t>(), c->tm[TM_BLK_OFF].as<uint32_t>(),
                                 ne_off, nj, job_off, slab_off, c->tm[TM_MULTI].as<uint32_t>(), c->tm[TM_NE_UNITS].as<uint32_t>(), c->tm[TM_NE_GEOM].as<int2>(), n_mt >> c->wsh, c->tm[TM_JOBS].as<TmJob>(), d_misc);
    {   // static work-balanced chunks of the job list; every workgroup of the walk should get several: a small load is cut finer
        DevBuf& pex = c->bt[BT_PEX];
        const uint64_t total_work = c->tm_np + (uint64_t)njobs * TM_JOB_W0;
        const uint64_t cw = total_work / ((uint64_t)c->n_cus * 14 * 6);
        const uint32_t chunk_work = (uint32_t)(cw < 256 ? 256 : (cw > TM_CHUNK_WORK ? TM_CHUNK_WORK : cw));
        if (pex.reserve(((size_t)njobs + 2) * 4) || c->tm[TM_CHUNKS].reserve(((size_t)(total_work / chunk_work) + 4) * 4)) return -1;
        hipcub::CountingInputIterator<uint32_t> iota(0);
        TmJobWork wf{c->tm[TM_JOBS].as<TmJob>()};
        hipcub::TransformInputIterator<uint32_t, TmJobWork, hipcub::CountingInputIterator<uint32_t>> it(iota, wf);
        SCAN_U32(it, pex.as<uint32_t>(), njobs);
        hipLaunchKernelGGL(k_tm_chunks, dim3((njobs + 255) / 256), dim3(256), 0, st, pex.as<uint32_t>(), njobs, chunk_work, c->tm[TM_CHUNKS].as<uint32_t>(), d_misc + 1);
    }
    LSG_HIP(hipMemcpyAsync(c->plan_misc, d_misc, 8, hipMemcpyDeviceToHost, st));
    c->d_plan_misc = d_misc;
    LSG_HIP(hipGetLastError());
    if (finish) LSG_HIP(hipStreamSynchronize(st));
    return 0;
}

static void plan_finish_tiles(lsg_ctx* c) {    // what the plan's tile-level half knows on the host
    const uint32_t* tot = c->plan1_tot;
    c->tm_njobs = tot[1];
    c->tm_n_ne = (uint32_t)((size_t)tot[0] * (size_t)c->n_ct); c->tm_n_multi = (uint32_t)((size_t)tot[3] * (size_t)c->n_ct); c->tm_n_slabs = tot[2];
}
static void plan_finish(lsg_ctx* c) {          // (after the stream plan_jobs ran on has been synchronised)
    plan_finish_tiles(c);
    c->tm_nchunks = c->plan_misc[1]; c->tm_n_wide = c->plan_misc[0];
    c->plan_n_ct = c->n_ct;
}

int ensure_plan(lsg_ctx* c) {
    if (!c->tm_valid) { set_error("lsg_pileup_count: no reads loaded"); return -2; }
    if (c->plan_n_ct == c->n_ct) return 0;
    c->plan_n_ct = 0;
    c->tm_njobs = c->tm_nchunks = c->tm_n_ne = c->tm_n_multi = c->tm_n_slabs = c->tm_n_wide = 0;
    if (c->tm_nblk == 0) { c->plan_n_ct = c->n_ct; return 0; }
    if (c->plan1_n_ct != c->n_ct) { if (int rc = plan_tiles(c, c->stream)) return rc; }
    if (int rc = plan_jobs(c, true)) return rc;
    plan_finish(c);
    return 0;
}

} // namespace lsg
