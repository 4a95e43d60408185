// Per-cell-type pileup base counting on CDNA4 (gfx950).
//
// Replaces, for every covered column at once:
//   split_bam's read routing            workflow/scripts/PreProcessing/SplitBamCellTypes.py:65-124
//   run_interval (pileup + counting)    workflow/scripts/SNVCalling/BaseCellCounter.py:182-320
//
// Three forms of the same count (identical rows: tests/test_paths_gpu.py), chosen per count by run_count:
//   scatter + sort   (a load's first count; > 2 cell types; reads dropped by max_depth)
//       unit  = (64-position tile of one contig, cell type); one lane per reference position.
//       entry = one read segment overlapping a unit, self-contained (barcode / strand key, line of its events), scattered into the
//               tile's static region (k_bin_segments).  Units deeper than CAPB entries are cut by barcode range into SLOTS of ~SUBT
//               entries (k_sort_deep), so every work item is bounded and the heavy tail (chrM, highly expressed genes) spreads
//               over the whole chip; counters are additive over disjoint barcode sets.  A slot's entries are grouped by barcode and
//               walked run by run: every entry is one 128-byte line of uint16 events (lane = position); per-symbol counts go to
//               LDS words with packed ds_adds; distinct-cell numbers are counts minus within-run duplicates.  Slots with <= CAPW
//               entries are processed by one wavefront each (k_pileup_wave), larger ones by 4 waves on run-aligned slices
//               (k_walk_block), partial sums of multi-slot units to slabs (k_finalize_multi).
//   tile index       (per load: every tile's entries sorted by barcode once, build_index)
//       k_resolve_agg + k_resolve write the walks' records in that order; k_wave_ix / k_cut + k_walk_block walk them.
//   tile-major store (per load and read filters: the admitted entries' events re-laid in index order, 8 entries to a transposed
//                     1 KB block, build_tm; after lsg_prepare_counts, or once a load has been counted three times)
//       k_tm_resolve (cell type per entry -> a meta word) + k_tm_walk (streams the blocks, both cell types in one pass).
//   No global atomics on the event path, integer arithmetic only (HBM- and issue-bound; no MFMA).
#include "lsg_ctx.h"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lsg {

constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;
constexpr uint32_t CB_MASK = 0x00FFFFFFu;
// Entries are stored PACKED in 8 bytes {cb | (events - 1) << 24 | forward << 30, index of the 128-byte line within the resident
// events}; unpack_entry() gives the working form used below:
// entry = {key, e, m, 0}: e = low 32 bits of the ADDRESS of the 128-byte line that holds the entry's 64-position tile slot
// (events are resident tile-aligned, layout.hip: lane = position within the tile, positions outside the segment are
// zero padding, so an entry is loaded as one whole line with no bounds and no first-lane arithmetic);
// m = [0..14] address bits 32..46, [15] the barcode run has exactly this one entry, [16..23] zero, [24..29] events - 1,
// [30] forward strand, [31] first entry of a barcode run (both run bits are set by the grouping step).  The layout lets
// the walk take "strand bit at packed position 14" straight from the upper half of m (bits 8..13 and 15 of that half are
// masked off again) and mask the address half with the one s_and that also drops bit 15.
constexpr uint32_t META_NEWRUN = 1u << 31, META_FWD = 1u << 30, META_SINGLE = 1u << 15;
__host__ __device__ __forceinline__ uint32_t meta_events(uint32_t m) { return ((m >> 24) & 63u) + 1u; }
#define LSG_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(LSG_AS3 const void*)p; }
constexpr int CAPW = 256;            // max entries of a wave-processed slot
constexpr int HW = 512;              // hash slots of the wave kernel
constexpr int CAPB = 2048;           // max entries staged at once by the block kernel
constexpr int HB = 4096;             // hash slots of the block kernel
constexpr int SUBT = 1024;           // target entries per barcode-range slot of a deep unit
constexpr int MAXSUB = 2048;         // max slots per unit
constexpr int NBUCKET = 256;         // coarse barcode buckets of the block kernel's fallback passes
constexpr int BLOCK_THREADS = 512;
constexpr int BLOCK_WAVES = BLOCK_THREADS / 64;
constexpr int WAVES_PER_BLOCK = 4;   // wave kernel
constexpr int ARENA = 256;           // rows reserved per wave per allocation: this many or a multiple (lsg_ctx::arena)
constexpr int QCHUNK = 64;          // slots dequeued at once by a wave
constexpr int FLUSH_EVERY = 63;      // packed LDS fields: bq 14 | fwd 6 | cnt 6 | dup 6 bits

// device scalars (uint64 each)
// The words kernels hammer while they run (work queues, row allocators) sit 128 bytes apart: atomics on one cache line serialise at
// ~90 per microsecond whichever of its words they name, and k_wave_ix alone issued 250 000 of them on the line all of these shared.
enum { SC_NNE = 2, SC_COLS = 8, SC_OVERFLOW = 9,
       SC_READS = 10, SC_SEGS = 11, SC_EVENTS = 12, SC_EV_WAVE = 13, SC_EV_DEEP = 14, SC_ROWS_DEEP = 15,
       SC_ROWS = 16, SC_NSMALL = 20, SC_NMULTI = 21, SC_NMULTI_SEL = 22, SC_NHUGE = 24,
       SC_NREST = 25, SC_ROWS_SRC = 28, SC_EV_SRC = 32, SC_NCHUNK = 36, SC_NENT = 38,
       SC_QSMALL = 48, SC_QBIG = 64, SC_QHUGE = 80, SC_QBIN0 = 96, SC_QBIN2 = 112, SC_QSORT = 128, SC_QREST = 144,
       SC_ROWALLOC = 160, SC_ROWALLOC_STRIDE = 16, SC_COUNT = SC_ROWALLOC + SC_ROWALLOC_STRIDE * LSG_MAX_CELLTYPES };   // *_SRC[4]: 0 wave, 1 walk_block, 2 huge, 3 finalize

struct CountArgs;
__device__ __forceinline__ uint4 unpack_entry(const CountArgs& a, uint2 p);

struct CountArgs {
    // reads
    int64_t n_reads, n_segs;
    const int32_t* read_tid; const uint16_t* read_flag; const uint8_t* read_mapq; const int32_t* read_cb;
    const uint32_t* seg_read; const int32_t* seg_start; const int32_t* seg_len; const int64_t* seg_ev_off;
    const uint16_t* events;
    // genome / barcodes
    const uint32_t* tile_base; const int64_t* contig_len; const uint8_t* const* ref_ptr;
    const uint8_t* celltype_of;
    int32_t n_contigs, n_cb, n_ct;
    uint32_t n_units;
    uint32_t tile_lo, tile_hi;          // counted tile range (lsg_set_region)
    // params
    int32_t min_bq, min_mq, min_dp, min_cc, ignore_orphans;
    uint32_t flag_exclude;
    // workspace
    uint32_t* read_key; uint32_t* unit_cnt; uint32_t* unit_off; uint32_t* unit_cursor;   // dense per unit: entry region of buffer A
    uint64_t ent_half;                    // entries of buffer B (barcode-split deep units) start here
    const uint32_t* ct_rank; uint32_t ct_size[LSG_MAX_CELLTYPES];   // rank of a barcode within its cell type
    uint32_t* ne_units; uint32_t* ne_nslot; uint32_t* ne_slot_base; uint32_t* ne_acc; int2* ne_geom;
    uint64_t* ne_mask; uint32_t* ne_rowbase;
    uint32_t* slot_w; uint32_t* slot_cnt; uint32_t* slot_off;
    uint2* ent;                           // packed entries, see unpack_entry
    uint2* seg_info;                      // per segment {admission key, first tile of its contig} (k_seg_info)
    uint2* rec;                           // grouped 8-byte records {event byte offset lo, meta} of the block path, same indexing as ent
    uint32_t* slot_pex; uint32_t* chunk_start;   // wave kernel: work prefix over the small-slot list, first slot of every chunk
    uint32_t* slot_list; uint32_t* multi_list; uint32_t* macc; uint32_t* slices; uint32_t* huge_list; uint32_t* rest_list;
    uint32_t n_ne, n_slots, n_multi;
    uint32_t arena;                       // rows a wave reserves per allocation (multiple of 256)
    uint32_t zero_lo, zero_hi;            // address (e, m form) of a 128-byte line of zeros behind the resident events
    uint32_t presorted;                   // multi-slot units' records were written grouped by k_sort_deep (no k_group_block pass, any slot size)
    uint32_t two_ended;                   // <= 2 cell types: a tile's static entry region is filled from both ends, no counting pass
    uint32_t inline_seg_info;             // k_bin_segments computes the admission record itself (no k_seg_info launch)
    const uint32_t* tile_off; uint32_t* cur_lo; uint32_t* cur_hi;      // [n_tiles + 1] static region starts; cursors of this count
    const uint8_t* read_drop;             // reads the pileup's max_depth rule drops (layout.hip depth_cap_drops), or null
    // tile index (static per load): entries in (tile, barcode) order
    const uint32_t* ix0; const uint32_t* ix1; const uint32_t* ix2;      // cb | events-1 << 24 | fwd << 30 | run start << 31; line; flag12 | mapq << 12 | tile start << 20 | segment start << 21
    const uint32_t* ix_netile; const int32_t* ix_chunk; unsigned long long* ix_carry;
    uint64_t ix_n;
    uint32_t index_path;                  // this count runs on the tile index (k_resolve) instead of the scatter + sort + group
    uint64_t* ixb_key; uint32_t* ixb_read;      // index build only: the scatter also writes (tile << 24 | cb) and the read of every entry
    unsigned long long* scalars;
    uint32_t* rows[LSG_MAX_CELLTYPES];
    uint64_t row_cap;
};

// packed entry -> {barcode, line address lo, meta (address bits 32..46 | events - 1 | strand), 0}
__device__ __forceinline__ uint4 unpack_entry(const CountArgs& a, uint2 p) {
    const uint64_t addr = (uint64_t)(uintptr_t)a.events + ((uint64_t)p.y << 7);
    return make_uint4(p.x & CB_MASK, (uint32_t)addr, ((uint32_t)(addr >> 32) & 0x7fffu) | (p.x & 0x7f000000u), 0u);
}


// ------------------------------------------------------------------------------------------------
// Read admission = the union of the reference's filters on the count path:
//   pysam pileup flag_filter (UNMAP|SECONDARY|QCFAIL|DUP) and min_mapping_quality, ignore_orphans
//   (BaseCellCounter.py:191), is_supplementary (:249), CB tag present (:240-243), CB in barcodes.tsv
//   with a cell type (SplitBamCellTypes.py:83-90), MAPQ >= min_MQ (:110-113).
// key = cb | reverse<<24 | celltype<<28.
__global__ void k_read_key(CountArgs a) {
    unsigned long long n_ok = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        uint32_t key = KEY_INVALID;
        uint32_t flag = a.read_flag[r];
        int32_t cb = a.read_cb[r];
        int32_t tid = a.read_tid[r];
        bool ok = (flag & a.flag_exclude) == 0 && (int)a.read_mapq[r] >= a.min_mq && cb >= 0 && cb < a.n_cb &&
                  tid >= 0 && tid < a.n_contigs;
        if (ok && a.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) ok = false;
        if (ok && a.read_drop && a.read_drop[r]) ok = false;          // bam.pileup(..., max_depth): never entered the pileup buffer
        if (ok) {
            uint32_t ct = a.celltype_of[cb];
            if (ct < (uint32_t)a.n_ct) key = (uint32_t)cb | (((flag >> 4) & 1u) << 24) | (ct << 28);
        }
        a.read_key[r] = key;
        n_ok += key != KEY_INVALID;
    }
    __shared__ unsigned long long s_ok;
    if (threadIdx.x == 0) s_ok = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) n_ok += __shfl_down(n_ok, o);
    if ((threadIdx.x & 63) == 0 && n_ok) atomicAdd(&s_ok, n_ok);
    __syncthreads();
    if (threadIdx.x == 0 && s_ok) atomicAdd(&a.scalars[SC_READS], s_ok);       // one same-address global atomic per workgroup
}

__device__ __forceinline__ uint32_t sub_of(uint32_t cb, uint32_t nsub, uint32_t n_cb) {
    return (uint32_t)(((uint64_t)cb * nsub) / n_cb);
}

// Counting sort of (segment, tile) pairs over the segments, with the atomics aggregated per
// workgroup in an LDS hash.  The segments of a coordinate-sorted BAM arrive gene by gene, so consecutive
// batches of 256 segments hit the same few units: a workgroup dequeues BIN_SUPER consecutive batches and
// keeps accumulating (batch, 8-tile round) items in the hash until it is 3/4 full, then issues ONE global
// atomic per distinct unit for the whole chunk.  (A deep gene funnels thousands of batches into a few
// cache lines of unit_cnt / unit_cursor; same-line atomics serialise in L2, so their number is what counts.)
//   MODE 0: count entries per unit.  MODE 2: claim a range per unit, then replay the chunk's items and
//   scatter the packed entries (barcode | count | strand, line index) into buffer A.
constexpr int BIN_THREADS = 256;
constexpr int BIN_TPR = 8;             // tiles per segment handled per item
constexpr int BIN_H = 4096;            // LDS hash slots
constexpr int BIN_SUPER = 16;          // batches per dequeue
constexpr int BIN_MAXI = 32;           // items per chunk
constexpr uint32_t BIN_FILL = BIN_H * 5 / 8;

struct BinSeg { uint32_t key, tb, t0, rd; int32_t st, ln, ntile; int64_t evoff; };

// Per-segment admission record {key, first tile of the contig}: the three-level gather segment -> read -> contig tables is
// done ONCE per count by this streaming kernel; the binning passes (one for the counts, two for the scatter) then read 8
// coalesced bytes per segment instead of chasing it again with a workgroup's few waves.
__global__ void k_seg_info(CountArgs a) {
    unsigned long long n_seg = 0, n_ev = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.n_segs; s += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = a.seg_read[s];
        uint32_t key = a.read_key[r], tb = 0;
        if (key != KEY_INVALID) {
            const int32_t tid = a.read_tid[r];
            const int64_t st = a.seg_start[s], ln = a.seg_len[s];
            if (st < 0 || ln <= 0 || st + ln > a.contig_len[tid]) key = KEY_INVALID;   // malformed: never counted
            else { tb = a.tile_base[tid]; ++n_seg; n_ev += (unsigned long long)ln; }
        }
        a.seg_info[s] = make_uint2(key, tb);
    }
    __shared__ unsigned long long s_st[2];
    if (threadIdx.x == 0) { s_st[0] = 0; s_st[1] = 0; }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { n_seg += __shfl_down(n_seg, o); n_ev += __shfl_down(n_ev, o); }
    if ((threadIdx.x & 63) == 0 && n_seg) { atomicAdd(&s_st[0], n_seg); atomicAdd(&s_st[1], n_ev); }
    __syncthreads();
    if (threadIdx.x == 0 && s_st[0]) { atomicAdd(&a.scalars[SC_SEGS], s_st[0]); atomicAdd(&a.scalars[SC_EVENTS], s_st[1]); }
}

template <int MODE>
__device__ __forceinline__ BinSeg bin_load(const CountArgs& a, int64_t s) {
    BinSeg g; g.key = KEY_INVALID; g.tb = 0; g.t0 = 0; g.rd = 0; g.st = 0; g.ln = 0; g.ntile = 0; g.evoff = 0;
    if (s < a.n_segs) {
        g.st = a.seg_start[s];
        g.ln = a.seg_len[s];
        if (a.inline_seg_info) {
            // the admission record computed here instead of by k_seg_info: with one binning pass left (two-ended scatter) it is read
            // twice, and the gathers segment -> read -> contig table run along the coordinate-sorted reads
            const uint32_t r = a.seg_read[s];
            g.key = a.read_key[r];
            if (g.key != KEY_INVALID) {
                const int32_t tid = a.read_tid[r];
                if (g.st < 0 || g.ln <= 0 || (int64_t)g.st + g.ln > a.contig_len[tid]) g.key = KEY_INVALID;
                else g.tb = a.tile_base[tid];
            }
        } else {
            const uint2 info = a.seg_info[s];
            g.key = info.x; g.tb = info.y;
        }
        if (MODE == 2 && g.key != KEY_INVALID) { g.evoff = a.seg_ev_off[s]; if (a.ixb_read) g.rd = a.seg_read[s]; }
    }
    bool ok = g.key != KEY_INVALID;
    uint32_t t0 = g.tb + ((uint32_t)g.st >> 6);
    uint32_t t1 = g.tb + ((uint32_t)(g.st + g.ln - 1) >> 6);
    if (t0 < a.tile_lo) t0 = a.tile_lo;
    if (a.tile_hi == 0) ok = false; else if (t1 + 1 > a.tile_hi) t1 = a.tile_hi - 1;
    g.t0 = t0;
    g.ntile = (ok && t1 >= t0) ? (int)(t1 - t0 + 1) : 0;
    return g;
}

template <int MODE>
__global__ __launch_bounds__(BIN_THREADS) void k_bin_segments(CountArgs a) {
    __shared__ uint32_t hkey[BIN_H], hcnt[BIN_H];      // MODE 2: hcnt turns into the unit's write cursor after the flush
    __shared__ uint32_t s_newb[BIN_MAXI], s_ib[BIN_MAXI], s_ir[BIN_MAXI];
    __shared__ int s_maxb[BIN_MAXI];
    __shared__ uint32_t s_super;
    const int t = threadIdx.x, lane = t & 63;
    constexpr int HSHIFT = 32 - __builtin_ctz(BIN_H);
    for (int i = t; i < BIN_H; i += BIN_THREADS) { hkey[i] = KEY_INVALID; hcnt[i] = 0; }
    const int64_t n_batches = (a.n_segs + BIN_THREADS - 1) / BIN_THREADS;
    const int64_t n_super = (n_batches + BIN_SUPER - 1) / BIN_SUPER;
    unsigned long long* qhead = &a.scalars[MODE == 0 ? SC_QBIN0 : SC_QBIN2];
    unsigned long long st_seg = 0, st_ev = 0;
    for (bool first = true;; first = false) {
        __syncthreads();
        // every workgroup's first item is its own index: no storm of same-address atomics at launch (~90 per us serialise)
        if (t == 0) s_super = first ? blockIdx.x : (uint32_t)atomicAdd(qhead, 1ull) + gridDim.x;
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
                const BinSeg g = bin_load<MODE>(a, b * BIN_THREADS + t);
                if (MODE == 2 && r == 0 && g.key != KEY_INVALID) { ++st_seg; st_ev += (unsigned long long)g.ln; }      // k_seg_info's statistics when it does not run
                const uint32_t ct = g.key >> 28;
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
                            const uint32_t x = (g.t0 + (uint32_t)k) * (uint32_t)a.n_ct + ct;
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
            // ---- one global atomic per distinct unit of the chunk
            for (int i = t; i < BIN_H; i += BIN_THREADS) {
                const uint32_t cnt = hcnt[i];
                if (cnt) {
                    if (MODE == 0) { atomicAdd(&a.unit_cnt[hkey[i]], cnt); hkey[i] = KEY_INVALID; hcnt[i] = 0; }
                    else if (a.two_ended) {
                        // cell type 0 grows up from the start of the tile's region, cell type 1 down from its end: the two units of a
                        // tile need no sizes in advance (the region holds every entry the tile can ever get)
                        const uint32_t u = hkey[i], tile = u / (uint32_t)a.n_ct;
                        hcnt[i] = u == tile * (uint32_t)a.n_ct ? atomicAdd(&a.cur_lo[tile], cnt) : atomicSub(&a.cur_hi[tile], cnt) - cnt;
                    }
                    else hcnt[i] = atomicAdd(&a.unit_cursor[hkey[i]], cnt);       // first position of this workgroup's range in the unit
                }
            }
            __syncthreads();
            if (MODE == 2) {
                // ---- pass B: replay the items, write the entries
                for (int it = 0; it < ni; ++it) {
                    const int64_t bb = b0 + s_ib[it];
                    const int rr = (int)s_ir[it];
                    const BinSeg g = bin_load<MODE>(a, bb * BIN_THREADS + t);
                    const uint32_t ct = g.key >> 28;
#pragma unroll
                    for (int j = 0; j < BIN_TPR; ++j) {
                        const int k = rr * BIN_TPR + j;
                        if (k < g.ntile) {
                            const uint32_t tt = g.t0 + (uint32_t)k;
                            const uint32_t x = tt * (uint32_t)a.n_ct + ct;
                            uint32_t h = (x * 2654435761u) >> HSHIFT;
                            while (hkey[h] != x) h = (h + 1) & (BIN_H - 1);
                            const uint32_t pos = atomicAdd(&hcnt[h], 1u);
                            const int32_t tstart = (int32_t)((tt - g.tb) << 6);
                            const int32_t lo = g.st > tstart ? g.st : tstart;
                            const int32_t hi = g.st + g.ln < tstart + TILE_W ? g.st + g.ln : tstart + TILE_W;
                            // the line of this tile slot: the event of (lo) sits at lane (lo - tstart) of it
                            const uint64_t line = (uint64_t)(g.evoff + (lo - g.st) - (lo - tstart)) >> 6;      // 64 events = 128 bytes
                            a.ent[pos] = make_uint2((g.key & CB_MASK) | ((uint32_t)(hi - lo - 1) << 24) | (((g.key >> 24) & 1u) ? 0u : META_FWD), (uint32_t)line);
                            if (a.ixb_key) {          // index build: sort key and owner of the entry; bit 31 of the read word = first entry of its segment
                                a.ixb_key[pos] = ((uint64_t)tt << 24) | (g.key & CB_MASK);
                                a.ixb_read[pos] = g.rd | (lo == g.st ? 0x80000000u : 0u);
                            }
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
    if (MODE == 2 && a.inline_seg_info) {
        for (int o = 32; o > 0; o >>= 1) { st_seg += __shfl_down(st_seg, o); st_ev += __shfl_down(st_ev, o); }
        __shared__ unsigned long long s_st[2];
        __syncthreads();
        if (t == 0) { s_st[0] = 0; s_st[1] = 0; }
        __syncthreads();
        if (lane == 0 && st_seg) { atomicAdd(&s_st[0], st_seg); atomicAdd(&s_st[1], st_ev); }
        __syncthreads();
        if (t == 0 && s_st[0]) { atomicAdd(&a.scalars[SC_SEGS], s_st[0]); atomicAdd(&a.scalars[SC_EVENTS], s_st[1]); }
    }
}

// Static per load: how many entries a tile can ever hold = the (segment, tile) overlaps of the reads that carry a barcode and lie on
// their contig (what k_seg_info can admit under ANY parameters or barcode table).  Counted by the counting pass itself
// (k_bin_segments<0>, atomics aggregated per workgroup) over this parameter-free admission record, as ONE cell type.
__global__ void k_seg_info_static(CountArgs a) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.n_segs; s += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = a.seg_read[s];
        const int32_t tid = a.read_tid[r];
        uint32_t key = KEY_INVALID, tb = 0;
        if (tid >= 0 && tid < a.n_contigs && a.read_cb[r] >= 0) {
            const int64_t st = a.seg_start[s], ln = a.seg_len[s];
            if (!(st < 0 || ln <= 0 || st + ln > a.contig_len[tid])) { key = (uint32_t)a.read_cb[r] | (((uint32_t)a.read_flag[r] >> 4 & 1u) << 24); tb = a.tile_base[tid]; }
        }
        a.seg_info[s] = make_uint2(key, tb);
    }
}

// The units' sizes and places after a two-ended scatter: unit (tile, 0) = [region start, low cursor), unit (tile, 1) = [high cursor, region end).
__global__ void k_units_from_cursors(CountArgs a) {
    unsigned long long tot = 0;
    // grid-stride: a few thousand workgroups, each ending in ONE atomic on the shared total (a word takes ~90 atomics per microsecond)
    for (uint32_t t = a.tile_lo + blockIdx.x * blockDim.x + threadIdx.x; t < a.tile_hi; t += gridDim.x * blockDim.x) {
        const uint32_t off = a.tile_off[t], end = a.tile_off[t + 1], lo = a.cur_lo[t], hi = a.cur_hi[t];
        if (lo > hi) atomicExch(&a.scalars[SC_OVERFLOW], 2ull);          // cannot happen while the capacities bound the entries
        if (a.n_ct == 1) { a.unit_cnt[t] = lo - off; a.unit_off[t] = off; tot += lo - off; }
        else { a.unit_cnt[2 * t] = lo - off; a.unit_off[2 * t] = off; a.unit_cnt[2 * t + 1] = end - hi; a.unit_off[2 * t + 1] = hi; tot += (lo - off) + (end - hi); }
    }
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    __shared__ unsigned long long s_tot;
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&s_tot, tot);
    __syncthreads();
    if (threadIdx.x == 0 && s_tot) atomicAdd(&a.scalars[SC_NENT], s_tot);
}

__device__ __forceinline__ void unit_geometry(const CountArgs& a, uint32_t u, int& ct, int& tid, int32_t& tstart) {
    uint32_t tile = u / (uint32_t)a.n_ct;
    ct = (int)(u - tile * (uint32_t)a.n_ct);
    int lo = 0, hi = a.n_contigs;                 // largest tid with tile_base[tid] <= tile
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.tile_base[mid] <= tile) lo = mid; else hi = mid; }
    tid = lo;
    tstart = (int32_t)((tile - a.tile_base[tid]) << 6);
}

// per non-empty unit: slot plan + geometry
__global__ void k_unit_plan(CountArgs a) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > a.n_ne) return;
    if (w == a.n_ne) { a.ne_nslot[w] = 0; a.ne_acc[w] = 0; return; }
    uint32_t u = a.ne_units[w];
    uint32_t cnt = a.unit_cnt[u];
    uint32_t nslot = 1;
    if (cnt > (uint32_t)CAPB) {
        nslot = (cnt + SUBT - 1) / SUBT;
        if (nslot > (uint32_t)MAXSUB) nslot = MAXSUB;
        { int ct0 = (int)(u % (uint32_t)a.n_ct); if (nslot > a.ct_size[ct0]) nslot = a.ct_size[ct0]; }
        if (nslot < 1) nslot = 1;
    }
    a.ne_nslot[w] = nslot;
    a.ne_acc[w] = nslot > 1 ? nslot : 0u;          // scanned afterwards: first partial-sum slab of the unit
    if (nslot > 1) atomicAdd(&a.scalars[SC_NMULTI], 1ull);
    int ct, tid; int32_t tstart;
    unit_geometry(a, u, ct, tid, tstart);
    a.ne_geom[w] = make_int2(tstart, tid | (ct << 24));
}

__global__ void k_slot_init(CountArgs a) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n_ne) return;
    uint32_t u = a.ne_units[w];
    uint32_t base = a.ne_slot_base[w], nslot = a.ne_nslot[w];
    for (uint32_t j = 0; j < nslot; ++j) a.slot_w[base + j] = w;
    if (nslot == 1) { a.slot_cnt[base] = a.unit_cnt[u]; a.slot_off[base] = a.unit_off[u]; }   // multi-slot units: k_split_deep
}

// Deep units (more than CAPB entries) are cut into slots by barcode rank: counting sort of the unit's
// entries from buffer A into buffer B (same unit offset), one workgroup per unit, LDS histogram.
constexpr int NSLICE = 4;             // run-aligned slices of a block-path slot = waves of k_walk_block
constexpr int SPLIT_THREADS = 512;
__global__ __launch_bounds__(SPLIT_THREADS) void k_split_deep(CountArgs a) {
    __shared__ uint32_t hist[MAXSUB];
    __shared__ uint32_t wave_tot[SPLIT_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t k = blockIdx.x; k < a.n_multi; k += gridDim.x) {
        const uint32_t w = a.multi_list[k], u = a.ne_units[w];
        const uint32_t n = a.unit_cnt[u], src = a.unit_off[u], nsub = a.ne_nslot[w], base = a.ne_slot_base[w];
        const uint32_t ct = (uint32_t)a.ne_geom[w].y >> 24;
        const uint32_t ctn = a.ct_size[ct];
        __syncthreads();
        for (int i = t; i < MAXSUB; i += SPLIT_THREADS) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += SPLIT_THREADS)
            atomicAdd(&hist[sub_of(a.ct_rank[a.ent[src + i].x & CB_MASK], nsub, ctn)], 1u);
        __syncthreads();
        // exclusive scan over MAXSUB = 4 x SPLIT_THREADS counters
        constexpr int PER = MAXSUB / SPLIT_THREADS;
        uint32_t loc[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { loc[q] = hist[t * PER + q]; sum += loc[q]; }
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t excl = incl - sum;
        for (int q = 0; q < wv; ++q) excl += wave_tot[q];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t j = (uint32_t)(t * PER + q);
            if (j < nsub) { a.slot_cnt[base + j] = loc[q]; a.slot_off[base + j] = (uint32_t)(a.ent_half + src + excl); }
            hist[j] = excl;                        // becomes the slot's write cursor
            excl += loc[q];
        }
        __syncthreads();
        for (uint32_t i = t; i < n; i += SPLIT_THREADS) {
            const uint2 e = a.ent[src + i];
            const uint32_t pos = atomicAdd(&hist[sub_of(a.ct_rank[e.x & CB_MASK], nsub, ctn)], 1u);
            a.ent[a.ent_half + src + pos] = e;
        }
    }
}

// Deep units when a cell type's barcodes fit an LDS table (SORT_RMAX): ONE counting sort by exact barcode rank
// replaces the split-by-range + per-slot grouping pair.  The unit's entries leave this kernel as the walk's 8-byte
// records, grouped by barcode with the run-start flags set (a record is the first of its run iff its position is
// the start of its barcode's range), cut into slots of ~n/nsub entries and NSLICE run-aligned slices per slot
// straight from the prefix sums.  Slots can be of any size here: the walk reads records, it stages nothing.
constexpr int SORT_RMAX = 16384;
#ifndef LSG_SORT_THREADS
#define LSG_SORT_THREADS 1024
#endif
constexpr int SORT_THREADS = LSG_SORT_THREADS;
__device__ __forceinline__ uint32_t lower_bound_lds(const uint32_t* start, uint32_t lo, uint32_t hi, uint32_t R, uint32_t n, uint32_t target) {
    // smallest r in [lo, hi] with start(r) >= target, start(R) = n
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = mid < R ? start[mid] : n;
        if (v >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__global__ __launch_bounds__(SORT_THREADS) void k_sort_deep(CountArgs a, uint32_t r_cap) {
    extern __shared__ uint32_t sort_lds[];
    uint32_t* start = sort_lds;                  // [r_cap] exclusive prefix of the per-barcode counts
    uint32_t* cur = sort_lds + r_cap;            // [r_cap] histogram, then scatter cursor
    uint32_t* sb = cur + r_cap;                  // [MAXSUB + 1] first rank of every slot
    uint32_t* wave_tot = sb + MAXSUB + 1;        // [SORT_THREADS / 64]
    unsigned long long* s_nev = reinterpret_cast<unsigned long long*>(wave_tot + SORT_THREADS / 64 + ((MAXSUB + 1 + SORT_THREADS / 64) & 1));
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned long long nev = 0;
    // units differ in size by orders of magnitude: taken off a queue (first one = own index), not strided
    for (bool first = true;; first = false) {
        __syncthreads();
        if (t == 0) *reinterpret_cast<uint32_t*>(s_nev + 1) = first ? blockIdx.x : (uint32_t)atomicAdd(&a.scalars[SC_QSORT], 1ull) + gridDim.x;
        __syncthreads();
        const uint32_t k = *reinterpret_cast<uint32_t*>(s_nev + 1);
        if (k >= a.n_multi) break;
        const uint32_t w = a.multi_list[k], u = a.ne_units[w];
        const uint32_t n = a.unit_cnt[u], src = a.unit_off[u], nsub = a.ne_nslot[w], base = a.ne_slot_base[w];
        const uint32_t ct = (uint32_t)a.ne_geom[w].y >> 24;
        const uint32_t R = a.ct_size[ct];
        __syncthreads();
        for (uint32_t i = t; i < R; i += SORT_THREADS) cur[i] = 0;
        __syncthreads();
        // four entries per thread and round: entry load -> rank lookup -> LDS add is a chain of dependent latencies, and the
        // deepest unit of a sample (10^5 entries) is sorted by ONE workgroup: independent chains side by side shorten that tail
        for (uint32_t i0 = t; i0 < n; i0 += 4 * SORT_THREADS) {
            uint4 e[4]; uint32_t r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint32_t i = i0 + q * SORT_THREADS; e[q] = unpack_entry(a, a.ent[src + (i < n ? i : i0)]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = a.ct_rank[e[q].x & CB_MASK];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + q * SORT_THREADS < n) { atomicAdd(&cur[r[q]], 1u); nev += meta_events(e[q].z); }
        }
        __syncthreads();
        // exclusive scan of cur[0..R) -> start; every thread owns a contiguous chunk
        const uint32_t chunk = (R + SORT_THREADS - 1) / SORT_THREADS;
        const uint32_t c_lo = (uint32_t)t * chunk < R ? (uint32_t)t * chunk : R, c_hi = c_lo + chunk < R ? c_lo + chunk : R;
        uint32_t sum = 0;
        for (uint32_t i = c_lo; i < c_hi; ++i) sum += cur[i];
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int q = 0; q < wv; ++q) run += wave_tot[q];
        for (uint32_t i = c_lo; i < c_hi; ++i) { const uint32_t c = cur[i]; start[i] = run; cur[i] = run; run += c; }
        __syncthreads();
        // slots: slot j starts at the first barcode whose range begins at or after j * n / nsub
        for (uint32_t j = t; j <= nsub; j += SORT_THREADS)
            sb[j] = j == nsub ? R : lower_bound_lds(start, 0, R, R, n, (uint32_t)(((uint64_t)n * j) / nsub));
        __syncthreads();
        for (uint32_t j = t; j < nsub; j += SORT_THREADS) {
            const uint32_t r0 = sb[j], r1 = sb[j + 1];
            const uint32_t s0 = r0 < R ? start[r0] : n, s1 = r1 < R ? start[r1] : n;
            const uint32_t slot = base + j;
            a.slot_cnt[slot] = s1 - s0;
            a.slot_off[slot] = (uint32_t)(a.ent_half + src + s0);
            for (int q = 0; q <= NSLICE; ++q) {
                uint32_t b = 0;
                if (q == NSLICE) b = s1 - s0;
                else if (q > 0) {
                    const uint32_t r = lower_bound_lds(start, r0, r1, R, n, s0 + (uint32_t)(((uint64_t)(s1 - s0) * q) / NSLICE));
                    b = (r < R ? start[r] : n) - s0;
                    if (b > s1 - s0) b = s1 - s0;
                }
                a.slices[(uint64_t)slot * (NSLICE + 1) + q] = b;
            }
        }
        // scatter the records (cur = cursor; start stays put)
        for (uint32_t i0 = t; i0 < n; i0 += 4 * SORT_THREADS) {
            uint4 e[4]; uint32_t r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint32_t i = i0 + q * SORT_THREADS; e[q] = unpack_entry(a, a.ent[src + (i < n ? i : i0)]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = a.ct_rank[e[q].x & CB_MASK];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q * SORT_THREADS >= n) continue;
                const uint32_t pos = atomicAdd(&cur[r[q]], 1u);
                const uint32_t r_end = r[q] + 1 < R ? start[r[q] + 1] : n;
                const uint32_t run = pos != start[r[q]] ? 0u : (r_end - pos == 1u ? (META_NEWRUN | META_SINGLE) : META_NEWRUN);
                a.rec[a.ent_half + src + pos] = make_uint2(e[q].y, e[q].z | run);
            }
        }
    }
    // events k_walk_block will read (statistics)
    for (int o = 32; o > 0; o >>= 1) nev += __shfl_down(nev, o);
    __syncthreads();
    if (t == 0) *s_nev = 0;
    __syncthreads();
    if (lane == 0 && nev) atomicAdd(s_nev, nev);
    __syncthreads();
    if (t == 0 && *s_nev) { atomicAdd(&a.scalars[SC_EV_DEEP], *s_nev); atomicAdd(&a.scalars[SC_EV_SRC + 1], *s_nev); }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
template <bool BLOCK> __device__ __forceinline__ void group_sync() {
    if (BLOCK) __syncthreads(); else lds_fence();
}
__device__ __forceinline__ uint32_t hash_cb(uint32_t cb) { return cb * 2654435761u; }
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// One pileup entry at a lane's position (BaseCellCounter.py:258-279).  m = the entry's meta word (wave-uniform:
// META_NEWRUN starts a new barcode run, META_FWD is the strand), ev = the event (0 when the lane is outside the
// segment: the tile-aligned layout holds zero padding there), thr = 0x800 + min_bq: an event is counted iff its valid bit
// is set and its quality passes the gate, i.e. (ev & 0x8ff) >= thr.  pkl = LDS byte address of this lane's word in
// row 0 of the wave's packed counters (2048-byte aligned rows block, so the row offset is OR-ed in).
// mask: bit 0 = any symbol seen in this barcode run, bit 8 + class = that class seen.  Branch-free.
__device__ __forceinline__ void pile_add(uint32_t& mask, uint32_t& ncdup, uint32_t m, uint32_t ev, uint32_t thr, uint32_t pkl) {
    mask &= (m & META_NEWRUN) ? 0u : 0xFFFFFFFFu;
    const uint32_t vm = (uint32_t)((int32_t)(thr - 1u - (ev & 0x8ffu)) >> 31);      // all ones when counted
    const uint32_t sym8 = __builtin_amdgcn_ubfe(ev, 8, 4);                          // 8 + class for a valid event
    const uint32_t seen = __builtin_amdgcn_ubfe(mask, sym8, 1);
    const uint32_t cst = (1u << 20) | ((m & META_FWD) ? (1u << 14) : 0u);          // scalar: count + strand
    const uint32_t val = ((seen << 26) | ((ev & 0xffu) | cst)) & vm;
    // ds_add_u32 on the lane-private word of the symbol's row (value 0 when not counted)
    __hip_atomic_fetch_add((LSG_AS3 uint32_t*)(uintptr_t)(pkl | (ev & 0x700u)), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ncdup += mask & vm & 1u;
    mask |= vm & ((1u << sym8) | 1u);
}

// Per-lane (= per reference position) accumulators of one unit.  Distinct-cell numbers come from
// duplicates: within a barcode run, an entry whose symbol was already seen at this position is a
// duplicate; CC[sym] = BC[sym] - dup[sym], NC = sum(BC) - ncdup   (= len(set(...)),
// BaseCellCounter.py:283,292).
struct Acc {
    uint32_t bc[8], bq[8], bcf[8], dup[8], ncdup;
    uint32_t mask, npk, nev;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < 8; ++s) bc[s] = bq[s] = bcf[s] = dup[s] = 0;
        ncdup = mask = npk = nev = 0;
    }
    __device__ __forceinline__ void flush_pk(uint32_t* pk, int lane) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint32_t v = pk[s * 64 + lane];
            pk[s * 64 + lane] = 0;
            bq[s] += v & 0x3fffu; bcf[s] += (v >> 14) & 63u; bc[s] += (v >> 20) & 63u; dup[s] += v >> 26;
        }
        npk = 0;
    }
    __device__ __forceinline__ void new_run() { mask = 0; }
    __device__ __forceinline__ uint32_t BC(int s) const { return bc[s]; }
    __device__ __forceinline__ uint32_t BQ(int s) const { return bq[s]; }
    __device__ __forceinline__ uint32_t BCF(int s) const { return bcf[s]; }
    __device__ __forceinline__ uint32_t DUP(int s) const { return dup[s]; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
    __device__ __forceinline__ void add(uint32_t m, uint32_t ev, uint32_t thr, uint32_t pkl) { pile_add(mask, ncdup, m, ev, thr, pkl); ++npk; }
    // call before adding up to `next` more entries: keeps the packed 6-bit fields from overflowing
    __device__ __forceinline__ void reserve(uint32_t next, uint32_t* pk, int lane) {
        if (npk + next > (uint32_t)FLUSH_EVERY) flush_pk(pk, lane);
    }
    __device__ __forceinline__ void finish(uint32_t* pk, int lane) { flush_pk(pk, lane); }
};

// The block path's accumulator: only the run state lives in registers; the packed counters are flushed straight
// into the workgroup's LDS accumulators [NCTR][64] (shared by its waves, hence ds_add), which keeps the walk loop
// at ~55 VGPRs = 8 waves per SIMD, and the event loads in flight are what hides HBM latency there.
struct WalkAcc {
    uint32_t nc, mask, npk, tot;         // nc: barcode runs with a counted event at this lane; tot: counted events flushed so far
    uint32_t open;                       // wave-uniform: a run of several entries is open (mask == 0 whenever it is not)
    uint32_t* sink;                      // [NCTR][64] words in LDS
    __device__ __forceinline__ void init(uint32_t* s) { nc = mask = npk = tot = 0; open = 0; sink = s; }
    __device__ __forceinline__ void flush_pk(uint32_t* pk, int lane) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t v = pk[s * 64 + lane];
            pk[s * 64 + lane] = 0;
            if (v) {
                const uint32_t bc = (v >> 20) & 63u;
                tot += bc;
                atomicAdd(&sink[(17 + s) * 64 + lane], v & 0x3fffu); atomicAdd(&sink[(25 + s) * 64 + lane], (v >> 14) & 63u);
                atomicAdd(&sink[(9 + s) * 64 + lane], bc); atomicAdd(&sink[(1 + s) * 64 + lane], v >> 26);
            }
        }
        npk = 0;
    }
    __device__ __forceinline__ void close_run() { nc += mask & 1u; mask = 0; }
    // One record (m wave-uniform, held in an SGPR by the scalar record loads): every decision below is a SCALAR branch.
    // A run of ONE entry cannot hold a duplicate (7 vector operations), the first entry of a longer run has nothing to compare
    // with (9), the others carry the seen-symbol mask and the duplicate bit (12); a run of several entries is closed when the
    // next run starts (3).  Written as one asm block: left to the compiler the branches become per-lane selects plus register
    // shuffles between the unrolled copies and cost more than the branch-free form (pile_add, 15.75 per entry) they replace.
    //   vm   = all ones when the event is counted: valid bit set and quality >= min_bq   (thr1 = 0x800 + min_bq - 1)
    //   base = quality | count 1 | strand (cst: bit 20, bit 14 when forward); addr = this lane's word in the symbol's row
    __device__ __forceinline__ void add(uint32_t m, uint32_t ev, uint32_t thr, uint32_t pkl) {
        uint32_t t0, t1, vm, base, addr;
        open = (uint32_t)__builtin_amdgcn_readfirstlane((int)open);        // uniform by construction; says so to the compiler
        // base = quality | upper half of m: the strand bit lands on packed bit 14; what else the half carries (events - 1 in
        // bits 8..13, the run bit in 15) is cleared by k1 / k2, which also add the count (bit 20) and keep the duplicate bit (26)
        asm volatile(
            "v_and_b32 %[t0], 0x8ff, %[ev]\n\t"
            "v_sub_u32 %[t0], %[thr1], %[t0]\n\t"
            "v_ashrrev_i32 %[vm], 31, %[t0]\n\t"
            "v_or_b32_sdwa %[base], %[m], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
            "v_and_or_b32 %[addr], %[ev], %[c700], %[pkl]\n\t"
            "s_bitcmp0_b32 %[m], 31\n\t"
            "s_cbranch_scc1 3f\n\t"                                  // not the start of a run
            "s_cmp_eq_u32 %[open], 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "v_and_b32 %[t0], 1, %[mask]\n\t"                          // close the run of several entries before this one
            "v_add_u32 %[nc], %[nc], %[t0]\n\t"
            "v_mov_b32 %[mask], 0\n"
            "1:\n\t"
            "v_and_or_b32 %[t0], %[base], %[k1], %[k20]\n\t"           // (base & 0x40ff) | 1 << 20
            "v_and_b32 %[t0], %[vm], %[t0]\n\t"
            "ds_add_u32 %[addr], %[t0]\n\t"
            "s_bitcmp0_b32 %[m], 15\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_sub_u32 %[nc], %[nc], %[vm]\n\t"                        // a run of one entry: counted once if counted
            "s_mov_b32 %[open], 0\n\t"
            "s_branch 4f\n"
            "2:\n\t"
            "v_bfe_u32 %[t0], %[ev], 8, 4\n\t"                         // first entry of a longer run: mask = its symbol
            "v_lshl_or_b32 %[t0], 1, %[t0], 1\n\t"
            "v_and_b32 %[mask], %[vm], %[t0]\n\t"
            "s_mov_b32 %[open], 1\n\t"
            "s_branch 4f\n"
            "3:\n\t"
            "v_bfe_u32 %[t0], %[ev], 8, 4\n\t"                         // 8 + class
            "v_bfe_u32 %[t1], %[mask], %[t0], 1\n\t"                   // symbol already seen in this run: duplicate
            "v_and_or_b32 %[base], %[base], %[k1], %[k20]\n\t"
            "v_lshl_or_b32 %[t1], %[t1], 26, %[base]\n\t"
            "v_and_b32 %[t1], %[vm], %[t1]\n\t"
            "ds_add_u32 %[addr], %[t1]\n\t"
            "v_lshl_or_b32 %[t0], 1, %[t0], 1\n\t"
            "v_and_or_b32 %[mask], %[t0], %[vm], %[mask]\n"
            "4:"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [vm] "=&v"(vm), [base] "=&v"(base), [addr] "=&v"(addr),
              [mask] "+v"(mask), [nc] "+v"(nc), [open] "+s"(open)
            : [ev] "v"(ev), [m] "s"(m), [thr1] "s"(thr - 1u), [c700] "s"(0x700u), [k1] "s"(0x40ffu), [k20] "v"(1u << 20), [pkl] "v"(pkl)
            : "scc", "memory");
        ++npk;
    }
    __device__ __forceinline__ void reserve(uint32_t next, uint32_t* pk, int lane) {
        if (npk + next > (uint32_t)FLUSH_EVERY) flush_pk(pk, lane);
    }
    // duplicates within barcode runs = counted events - runs that counted one (NC = sum(BC) - this, see Acc)
    __device__ __forceinline__ void finish(uint32_t* pk, int lane) {
        if (open) { close_run(); open = 0; }
        flush_pk(pk, lane);
        atomicAdd(&sink[lane], tot - nc);
    }
};

// a unit's finished counters read straight from an LDS accumulator block [NCTR][64] (no register copy)
struct LdsCounters {
    const uint32_t* s; int lane;
    __device__ __forceinline__ uint32_t BC(int k) const { return s[(9 + k) * 64 + lane]; }
    __device__ __forceinline__ uint32_t BQ(int k) const { return s[(17 + k) * 64 + lane]; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return s[(25 + k) * 64 + lane]; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return s[(1 + k) * 64 + lane]; }
    __device__ __forceinline__ uint32_t NCDUP() const { return s[lane]; }
};

// Event load of one entry (es, ms wave-uniform; lane2 = 2 * lane): the entry's whole 128-byte line through a raw buffer
// descriptor {line address, 128 bytes}.  Lanes outside the segment read the layout's zero padding (= not countable).
__device__ __forceinline__ uint32_t load_event(uint32_t es, uint32_t ms, uint32_t lane2) {
    char* base = reinterpret_cast<char*>((uintptr_t)((((uint64_t)(ms & 0x7fffu)) << 32) | es));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 128, 0x00020000);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)lane2, 0, 0);
}

__device__ __forceinline__ uint32_t bq_threshold(const CountArgs& a) {
    const int q = a.min_bq < 0 ? 0 : (a.min_bq > 256 ? 256 : a.min_bq);
    return 0x800u + (uint32_t)q;
}
// Up to 64 records held one per lane (e, m; lanes past the last record hold 0 = an empty entry): groups of 8 with the
// event loads of group g+1 issued before group g is consumed.
__device__ __forceinline__ void issue8r(uint32_t e, uint32_t m, int l0, uint32_t lane2, uint32_t (&ms)[8], uint32_t (&ev)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { ms[u] = rl(m, l0 + u); ev[u] = load_event(rl(e, l0 + u), ms[u], lane2); }
}
// ALONE: every entry is a barcode run of its own (the caller checked that no barcode repeats), so there is nothing a
// duplicate could be compared with: count + quality + strand only
template <bool ALONE>
__device__ __forceinline__ void consume8r(Acc& acc, const uint32_t (&ms)[8], const uint32_t (&ev)[8], uint32_t thr, uint32_t* pk, int lane) {
    acc.reserve(8, pk, lane);
    const uint32_t pkl = lds_addr(pk + lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (ALONE) {
            const uint32_t vm = (uint32_t)((int32_t)(thr - 1u - (ev[u] & 0x8ffu)) >> 31);
            const uint32_t cst = (1u << 20) | ((ms[u] & META_FWD) ? (1u << 14) : 0u);
            __hip_atomic_fetch_add((LSG_AS3 uint32_t*)(uintptr_t)(pkl | (ev[u] & 0x700u)), ((ev[u] & 0xffu) | cst) & vm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++acc.npk;
        } else {
            acc.add(ms[u], ev[u], thr, pkl);
        }
    }
}
template <bool ALONE = false>
__device__ __forceinline__ void walk_regs(Acc& acc, uint32_t e, uint32_t m, int nb, uint32_t thr, uint32_t* pk, int lane) {
    const uint32_t lane2 = 2u * (uint32_t)lane;
    const int ng = (nb + 7) >> 3;
    if (ng <= 0) return;
    uint32_t msA[8], evA[8], msB[8], evB[8];
    issue8r(e, m, 0, lane2, msA, evA);
    int g = 0;
    while (true) {
        if (g + 1 < ng) issue8r(e, m, (g + 1) * 8, lane2, msB, evB);
        consume8r<ALONE>(acc, msA, evA, thr, pk, lane);
        if (++g >= ng) break;
        if (g + 1 < ng) issue8r(e, m, (g + 1) * 8, lane2, msA, evA);
        consume8r<ALONE>(acc, msB, evB, thr, pk, lane);
        if (++g >= ng) break;
    }
}
__device__ __forceinline__ void walk(const CountArgs& a, Acc& acc, const uint32_t* gev, const uint32_t* gmeta,
                                     int j0, int j1, uint32_t* pk, int lane) {
    const uint32_t thr = bq_threshold(a);
    for (int jb = j0; jb < j1; jb += 64) {
        const int nb = j1 - jb < 64 ? j1 - jb : 64;
        uint32_t e = a.zero_lo, m = a.zero_hi;                      // lanes past the last record: a line of zeros
        if (lane < nb) { e = gev[jb + lane]; m = gmeta[jb + lane]; acc.nev += meta_events(m); }
        walk_regs(acc, e, m, nb, thr, pk, lane);
    }
}

// Group n entries (global SoA arrays at src) by barcode into the LDS arrays gkey/gev/gmeta:
// entries with equal barcodes become adjacent.  T threads cooperate (T = 64: one wave, fences
// only; T = BLOCK_THREADS: __syncthreads).  When `filter` is set only entries whose barcode bucket
// (cb >> shift) lies in [b_lo, b_hi) are taken (fallback passes).
template <bool BLOCK, int H, int CAP, bool TO_GLOBAL = false>
__device__ __forceinline__ int group_by_cb(const CountArgs& a, uint32_t src, int n, uint32_t* gkey, uint32_t* gev, uint32_t* gmeta,
                                           uint32_t* tkey, uint32_t* tcnt, int t, uint32_t* wave_tot) {
    constexpr int T = BLOCK ? BLOCK_THREADS : 64;
    constexpr int RMAX = (CAP + T - 1) / T;
    for (int i = t; i < H; i += T) { tkey[i] = KEY_INVALID; tcnt[i] = 0; }
    group_sync<BLOCK>();
    uint32_t ek[RMAX], ee[RMAX], em[RMAX], hs[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        int i = t + r * T;
        hs[r] = 0; ek[r] = KEY_INVALID; ee[r] = 0; em[r] = 0;
        if (i < n) { const uint4 v = unpack_entry(a, a.ent[src + i]); ek[r] = v.x; ee[r] = v.y; em[r] = v.z; }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (t + r * T < n) {
            uint32_t cb = ek[r] & CB_MASK;
            uint32_t h = hash_cb(cb) >> (32 - __builtin_ctz(H));
            while (true) {
                uint32_t prev = atomicCAS(&tkey[h], KEY_INVALID, cb);
                if (prev == KEY_INVALID || prev == cb) break;
                h = (h + 1) & (H - 1);
            }
            uint32_t rank = atomicAdd(&tcnt[h], 1u);
            hs[r] = h | (rank << 16);
        }
    }
    group_sync<BLOCK>();
    // exclusive scan of tcnt over H slots; each thread owns H/T consecutive slots
    constexpr int PER = H / T;
    uint32_t loc[PER]; uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = tcnt[t * PER + q]; sum += loc[q]; }
    uint32_t incl = sum;
    int lane = t & 63;
    for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    uint32_t excl = incl - sum;
    if (BLOCK) {
        int w = t >> 6;
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        uint32_t add = 0;
        for (int q = 0; q < w; ++q) add += wave_tot[q];
        excl += add;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) { tcnt[t * PER + q] = excl; excl += loc[q]; }
    group_sync<BLOCK>();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (t + r * T < n) {
            uint32_t p = tcnt[hs[r] & 0xffffu] + (hs[r] >> 16);
            hs[r] = p;
            if (TO_GLOBAL) {
                gkey[p] = ek[r] & CB_MASK;
            } else {
                gkey[p] = ek[r]; gev[p] = ee[r]; gmeta[p] = em[r];
            }
        }
    }
    group_sync<BLOCK>();
    // mark the first entry of every barcode run (the walk resets its per-run symbol mask there)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (t + r * T < n) {
            const uint32_t p = hs[r];
            const bool first = p == 0 || (gkey[p - 1] & CB_MASK) != (ek[r] & CB_MASK);
            if (TO_GLOBAL) {                                                                       // the walk's 8-byte records
                const bool last = p + 1 == (uint32_t)n || (gkey[p + 1] & CB_MASK) != (ek[r] & CB_MASK);
                a.rec[src + p] = make_uint2(ee[r], em[r] | (first ? (last ? (META_NEWRUN | META_SINGLE) : META_NEWRUN) : 0u));
            }
            else if (first) atomicOr(&gmeta[p], META_NEWRUN);
        }
    }
    group_sync<BLOCK>();
    return n;
}

// wave-private bookkeeping (LDS): row arenas and exact counters
struct WaveBook {
    uint32_t arena_next[LSG_MAX_CELLTYPES], arena_end[LSG_MAX_CELLTYPES], rows_true[LSG_MAX_CELLTYPES];
    uint32_t cols, rows_deep, rows_src, src;
    unsigned long long nev;
};
__device__ __forceinline__ void book_init(WaveBook& b, int lane) {
    if (lane < LSG_MAX_CELLTYPES) { b.arena_next[lane] = 0; b.arena_end[lane] = 0; b.rows_true[lane] = 0; }
    if (lane == 0) { b.cols = 0; b.rows_deep = 0; b.rows_src = 0; b.src = 0; }
}
__device__ __forceinline__ void book_flush(const CountArgs& a, WaveBook& b, int lane) {
    lds_fence();
    if (lane < a.n_ct && b.rows_true[lane]) atomicAdd(&a.scalars[SC_ROWS + lane], (unsigned long long)b.rows_true[lane]);
    if (lane == 0 && b.cols) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)b.cols);
    if (lane == 0 && b.rows_deep) atomicAdd(&a.scalars[SC_ROWS_DEEP], (unsigned long long)b.rows_deep);
    if (lane == 0 && b.rows_src) atomicAdd(&a.scalars[SC_ROWS_SRC + b.src], (unsigned long long)b.rows_src);
}

// Gates + row emission for one unit by one wave.  Gates: BaseCellCounter.py:211 (ref != N), :282
// (count >= MIN_COV), :294 (NC >= MIN_CC); position 0 of a contig is never visited (:86).
// bk != nullptr: rows come from the wave's arena; nullptr: one exact global atomic.
// NARROW (the caller guarantees every value < 2^16, and that its arenas are whole 64-row blocks of its own): the unit's rows are
// written as 16-bit planes into the first half of their blocks (same row numbering, half the bytes moved here and in the call
// stage's gather); bit 31 of the unit's row base says so (ROW_NARROW).
template <class CNT, bool NARROW = false>
__device__ __forceinline__ void emit_unit(const CountArgs& a, const CNT& acc, uint32_t w, int ct, int tid, int32_t tstart, int lane,
                                          WaveBook* bk, bool deep, int ref_prefetched = -1, int arena_slot = -1) {
    const int as = arena_slot < 0 ? ct : arena_slot;          // a wave that writes narrow AND wide rows keeps an arena per format (whole blocks of one format)
    uint32_t dp = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) dp += acc.BC(s);
    const uint32_t nc = dp - acc.NCDUP();
    int64_t pos = (int64_t)tstart + lane;
    bool valid = pos >= 1 && pos < a.contig_len[tid];
    uint8_t refb = 'N';
    if (ref_prefetched >= 0) refb = (uint8_t)ref_prefetched;
    else if (valid) refb = a.ref_ptr[tid][pos];
    unsigned long long colm = __ballot(valid && dp > 0);
    bool emit = valid && dp > 0 && (int)dp >= a.min_dp && (int)nc >= a.min_cc && refb != 'N';
    unsigned long long em = __ballot(emit);
    const uint32_t k = (uint32_t)__popcll(em);
    uint32_t base = 0;
    if (lane == 0) {
        if (bk) {
            bk->cols += (uint32_t)__popcll(colm);
            if (k) {
                uint32_t nx = bk->arena_next[as];
                if (nx + k > bk->arena_end[as]) {
                    nx = (uint32_t)atomicAdd(&a.scalars[SC_ROWALLOC + SC_ROWALLOC_STRIDE * ct], (unsigned long long)a.arena);
                    bk->arena_end[as] = nx + a.arena;
                }
                base = nx; bk->arena_next[as] = nx + k; bk->rows_true[ct] += k;
                if (deep) bk->rows_deep += k;
                bk->rows_src += k;
            }
        } else {
            if (colm) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)__popcll(colm));
            if (k) {
                base = (uint32_t)atomicAdd(&a.scalars[SC_ROWALLOC + SC_ROWALLOC_STRIDE * ct], (unsigned long long)k);
                atomicAdd(&a.scalars[SC_ROWS + ct], (unsigned long long)k);
                if (deep) atomicAdd(&a.scalars[SC_ROWS_DEEP], (unsigned long long)k);
                atomicAdd(&a.scalars[SC_ROWS_SRC + 3], (unsigned long long)k);
            }
        }
        a.ne_mask[w] = em;
        a.ne_rowbase[w] = base | (NARROW ? ROW_NARROW : 0u);
    }
    base = rl(base, 0);
    if (!em) return;
    if ((uint64_t)base + k > a.row_cap) {
        if (lane == 0) atomicExch(&a.scalars[SC_OVERFLOW], 1ull);
        return;
    }
    if (!emit) return;
    // the unit's rows lie in at most two consecutive 64-row blocks: ONE descriptor over those blocks (scalar registers), the
    // lane adds its row's offset inside them and moves four planes per 16-byte store (quad = scalar offset of 1024 bytes)
    const uint32_t row = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
    const uint32_t b0 = base >> 6;
    const uint32_t off16 = (((row >> 6) - b0) * (uint32_t)ROW_BLOCK_WORDS + (row & 63u) * 4u) * 4u;
    const uint64_t first = (uint64_t)(uintptr_t)a.rows[ct] + (uint64_t)b0 * (ROW_BLOCK_WORDS * 4ull);
    const uint64_t p0 = ((uint64_t)rl((uint32_t)(first >> 32), 0) << 32) | rl((uint32_t)first, 0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)p0), 0, (int)(2 * ROW_BLOCK_WORDS * 4), 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto plane = [&](int p) -> uint32_t {                             // the row in plane order: DP, NC, CC[8], BC[8], BQ[8], BCf[8], 0, 0
        if (p == 0) return dp;
        if (p == 1) return nc;
        if (p < 10) return acc.BC(p - 2) - acc.DUP(p - 2);
        if (p < 18) return acc.BC(p - 10);
        if (p < 26) return acc.BQ(p - 18);
        if (p < 34) return acc.BCF(p - 26);
        return 0u;
    };
    if (NARROW) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const uint32_t off8 = ((row >> 6) - b0) * (uint32_t)ROW_BLOCK_WORDS * 4u + (row & 63u) * 8u;
#pragma unroll
        for (int q = 0; q < ROW_QUADS; ++q) {
            u32x2 v; v.x = plane(4 * q) | (plane(4 * q + 1) << 16); v.y = plane(4 * q + 2) | (plane(4 * q + 3) << 16);
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off8, q * 512, 0);
            asm volatile("s_nop 3");                                       // as below
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < ROW_QUADS; ++q) {
        u32x4 v; v.x = plane(4 * q); v.y = plane(4 * q + 1); v.z = plane(4 * q + 2); v.w = plane(4 * q + 3);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off16, q * 1024, 0);
        // gfx950 reads the data registers of a 128-bit buffer store late; the compiler guards that with wait states only when the
        // store has no SGPR offset, and reuses the registers for the next quad at once.  Measured without the s_nop: word 0 of a quad
        // took the next quad's value in ~1e-4 of the rows (tools/row_diff.py).  The barrier keeps the next quad's moves behind the nop.
        asm volatile("s_nop 3");
        __builtin_amdgcn_sched_barrier(0);                             // also: one quad's values live at a time (counters read from LDS stay there until needed)
    }
}

// ------------------------------------------------------------------------------------------------
// Wave kernel: each wavefront pulls QCHUNK single-slot units with <= CAPW entries at a time.
struct alignas(2048) WaveLds {
    uint32_t tkey[HW], tcnt[HW];          // pk (8*64 words, 2048-byte aligned) aliases tkey after grouping
    uint32_t gkey[CAPW], gev[CAPW], gmeta[CAPW];
    WaveBook book;
};

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void k_pileup_wave(CountArgs a) {
    __shared__ WaveLds lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    WaveLds& L = lds_all[threadIdx.x >> 6];
    uint32_t* pk = L.tkey;
    static_assert(HW >= 8 * 64, "pk must fit in the hash key array");
    book_init(L.book, lane);
    for (int i = lane; i < 8 * 64; i += 64) pk[i] = 0;
    const uint32_t n_chunks = (uint32_t)a.scalars[SC_NCHUNK];
    unsigned long long nev_total = 0;
    const uint32_t n_waves_all = gridDim.x * WAVES_PER_BLOCK;
    for (bool first = true;; first = false) {
        // chunks hold about the same WORK (entries + a per-slot constant), not the same number of slots; a wave's first
        // chunk is its own index, later ones come off the queue
        uint32_t ck = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
        if (!first) {
            if (lane == 0) ck = (uint32_t)atomicAdd(&a.scalars[SC_QSMALL], 1ull) + n_waves_all;
            ck = rl(ck, 0);
        }
        if (ck >= n_chunks) break;
        const uint32_t q0 = a.chunk_start[ck];
        const int nq = (int)(a.chunk_start[ck + 1] - q0);
        // lanes 0..nq-1 fetch their slot's descriptor
        uint32_t s_w = 0, s_off = 0, s_cnt = 0; int2 s_geom = make_int2(0, 0);
        if (lane < nq) {
            uint32_t s = a.slot_list[q0 + lane];
            s_w = a.slot_w[s]; s_off = a.slot_off[s]; s_cnt = a.slot_cnt[s];
            s_geom = a.ne_geom[s_w];
        }
        // slot 0's entries (one per lane) when it is a one-batch slot; later slots are prefetched one slot ahead
        uint4 cur = make_uint4(KEY_INVALID, 0u, 0u, 0u);
        auto fetch = [&](uint32_t off) -> uint4 { return unpack_entry(a, a.ent[off + lane]); };
        {
            const int n0 = (int)rl(s_cnt, 0);
            if (n0 <= 64 && lane < n0) cur = fetch(rl(s_off, 0));
        }
        for (int qi = 0; qi < nq; ++qi) {
            const uint32_t w = rl(s_w, qi), src = rl(s_off, qi);
            const int n = (int)rl(s_cnt, qi);
            const int32_t tstart = (int32_t)rl((uint32_t)s_geom.x, qi);
            const uint32_t g = rl((uint32_t)s_geom.y, qi);
            const int tid = (int)(g & 0xffffffu), ct = (int)(g >> 24);
            // memory the slot will want later: the next slot's entries, this slot's reference bases
            uint4 nxt = make_uint4(KEY_INVALID, 0u, 0u, 0u);
            if (qi + 1 < nq) {
                const int nn = (int)rl(s_cnt, qi + 1);
                if (nn <= 64 && lane < nn) nxt = fetch(rl(s_off, qi + 1));
            }
            int refb = 'N';
            { const int64_t pos = (int64_t)tstart + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            Acc acc; acc.init();
            bool general = n > 64;
            if (!general) {
                // one batch: if no barcode occurs twice every entry is its own run and no grouping is needed
                L.tcnt[lane] = KEY_INVALID; L.tcnt[lane + 64] = KEY_INVALID;
                lds_fence();
                bool dup = false;
                if (lane < n) {
                    const uint32_t cb = cur.x & CB_MASK;
                    uint32_t h = hash_cb(cb) >> 25;
                    while (true) {
                        const uint32_t prev = atomicCAS(&L.tcnt[h], KEY_INVALID, cb);
                        if (prev == KEY_INVALID) break;
                        if (prev == cb) { dup = true; break; }
                        h = (h + 1) & 127u;
                    }
                }
                general = __ballot(dup) != 0ull;
                if (!general) {
                    const uint32_t m = lane < n ? (cur.z | META_NEWRUN) : a.zero_hi;      // lanes past the last entry: a line of zeros
                    if (lane < n) acc.nev += meta_events(m);
                    walk_regs<true>(acc, lane < n ? cur.y : a.zero_lo, m, n, bq_threshold(a), pk, lane);
                }
            }
            if (general) {
                lds_fence();
                group_by_cb<false, HW, CAPW>(a, src, n, L.gkey, L.gev, L.gmeta, L.tkey, L.tcnt, lane, nullptr);
                for (int i = lane; i < 8 * 64; i += 64) pk[i] = 0;
                walk(a, acc, L.gev, L.gmeta, 0, n, pk, lane);
            }
            acc.finish(pk, lane);                                     // leaves the packed counters zeroed for the next slot
            nev_total += acc.nev;
            emit_unit(a, acc, w, ct, tid, tstart, lane, &L.book, false, refb);
            cur = nxt;
        }
    }
    // exact counters: summed over the workgroup's waves first (all waves of the grid finish together, and
    // same-line global atomics serialise: one set per workgroup instead of one per wave)
    for (int o = 32; o > 0; o >>= 1) nev_total += __shfl_down(nev_total, o);
    if (lane == 0) L.book.nev = nev_total;
    lds_fence();
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned long long nev = 0; uint32_t rt = 0, cols = 0, rdeep = 0, rsrc = 0;
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
            const WaveBook& b = lds_all[w].book;
            if (lane < a.n_ct) rt += b.rows_true[lane];
            cols += b.cols; rdeep += b.rows_deep; rsrc += b.rows_src; nev += b.nev;
        }
        if (lane < a.n_ct && rt) atomicAdd(&a.scalars[SC_ROWS + lane], (unsigned long long)rt);
        if (lane == 0) {
            if (cols) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)cols);
            if (rdeep) atomicAdd(&a.scalars[SC_ROWS_DEEP], (unsigned long long)rdeep);
            if (rsrc) atomicAdd(&a.scalars[SC_ROWS_SRC + 0], (unsigned long long)rsrc);
            if (nev) { atomicAdd(&a.scalars[SC_EV_WAVE], nev); atomicAdd(&a.scalars[SC_EV_SRC + 0], nev); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave kernel of the tile index path.  The records arrive grouped and carry their run flags, so nothing is staged or hashed: the
// wave's LDS is its packed counters alone, in two planes wide enough for a whole small unit (<= CAPW = 256 entries, 256 x 255 < 2^16)
// without a flush:   plane 0: quality sum [0..15] | forward count [16..31]      plane 1: count [0..15] | duplicates [16..31]
// 4 KB of LDS per wave and the run state in ~60 VGPRs instead of 32 KB per workgroup and 143: 8 waves per SIMD instead of 3.
// Per entry the lanes that count the event are selected with EXEC (v_cmpx) instead of a mask word, which takes the "& vm" operations
// out of every path:  run of one entry 5 vector operations, first entry of a longer run 6 (+3 when a run is closed), others 9.
template <int FWD_BIT>                  // where plane 0 counts the forward strand: above the quality sum's field
struct PlaneAcc {
    uint32_t nc, mask, open;             // nc: barcode runs with a counted event at this lane; open: wave-uniform, a run of several entries is open
    __device__ __forceinline__ void init() { nc = mask = 0; open = 0; }
    // m: the record's meta word (SGPR), ev: the lane's event, thr = 0x800 + min_bq, pkl: LDS byte address of this lane's word in row 0
    // of plane 0 (4096-byte aligned block: the row and the plane are OR-ed / offset in), one: a VGPR holding 1
    __device__ __forceinline__ void add(uint32_t m, uint32_t ev, uint32_t thr, uint32_t pkl, uint32_t one) {
        uint32_t t0, t1, addr, lo;
        unsigned long long sv;
        open = (uint32_t)__builtin_amdgcn_readfirstlane((int)open);
        const uint32_t sfwd = ((m >> 30) & 1u) << FWD_BIT;                // forward strand: one in the upper field of plane 0
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "v_and_b32 %[t0], 0x8ff, %[ev]\n\t"
            "v_and_or_b32 %[addr], %[ev], %[c700], %[pkl]\n\t"
            "v_or_b32_sdwa %[lo], %[sfwd], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
            "s_bitcmp0_b32 %[m], 31\n\t"
            "s_cbranch_scc1 3f\n\t"                                  // not the start of a run
            "s_cmp_eq_u32 %[open], 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "v_and_b32 %[t1], 1, %[mask]\n\t"                          // close the run of several entries before this one (every lane)
            "v_add_u32 %[nc], %[nc], %[t1]\n\t"
            "v_mov_b32 %[mask], 0\n"
            "1:\n\t"
            "v_cmpx_le_u32 vcc, %[thr], %[t0]\n\t"                     // EXEC = the lanes that count this event
            "ds_add_u32 %[addr], %[lo]\n\t"
            "ds_add_u32 %[addr], %[one] offset:2048\n\t"
            "s_bitcmp0_b32 %[m], 15\n\t"
            "s_cbranch_scc1 2f\n\t"
            "v_add_u32 %[nc], 1, %[nc]\n\t"                            // a run of one entry: counted once if counted
            "s_mov_b32 %[open], 0\n\t"
            "s_branch 4f\n"
            "2:\n\t"
            "v_bfe_u32 %[t1], %[ev], 8, 4\n\t"                         // first entry of a longer run: mask = its symbol (8 + class) and bit 0
            "v_lshl_or_b32 %[mask], %[one], %[t1], %[one]\n\t"
            "s_mov_b32 %[open], 1\n\t"
            "s_branch 4f\n"
            "3:\n\t"
            "v_cmpx_le_u32 vcc, %[thr], %[t0]\n\t"
            "v_bfe_u32 %[t1], %[ev], 8, 4\n\t"
            "v_bfe_u32 %[t0], %[mask], %[t1], 1\n\t"                   // symbol already seen in this run: duplicate
            "v_lshl_or_b32 %[t0], %[t0], 16, %[one]\n\t"
            "ds_add_u32 %[addr], %[lo]\n\t"
            "ds_add_u32 %[addr], %[t0] offset:2048\n\t"
            "v_lshl_or_b32 %[t1], %[one], %[t1], %[one]\n\t"
            "v_or_b32 %[mask], %[mask], %[t1]\n"
            "4:\n\t"
            "s_mov_b64 exec, %[sv]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [addr] "=&v"(addr), [lo] "=&v"(lo), [sv] "=&s"(sv),
              [mask] "+v"(mask), [nc] "+v"(nc), [open] "+s"(open)
            : [ev] "v"(ev), [m] "s"(m), [thr] "s"(thr), [c700] "s"(0x700u), [sfwd] "s"(sfwd), [one] "v"(one), [pkl] "v"(pkl)
            : "scc", "vcc", "memory");
    }
    __device__ __forceinline__ void finish() { if (open) { nc += mask & 1u; mask = 0; open = 0; } }
};
typedef PlaneAcc<16> IxAcc;              // small units: 256 x 255 < 2^16
// a small unit's finished counters read from the two planes
struct IxCounters {
    const uint32_t* pk; int lane; uint32_t ncdup;
    __device__ __forceinline__ uint32_t BC(int k) const { return pk[512 + k * 64 + lane] & 0xffffu; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return pk[512 + k * 64 + lane] >> 16; }
    __device__ __forceinline__ uint32_t BQ(int k) const { return pk[k * 64 + lane] & 0xffffu; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return pk[k * 64 + lane] >> 16; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};
__device__ __forceinline__ void issue8x(uint32_t e, uint32_t m, int l0, uint32_t lane2, uint32_t (&ev)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ev[u] = load_event(rl(e, l0 + u), rl(m, l0 + u), lane2);
}
__device__ __forceinline__ void walk_regs_ix(IxAcc& acc, uint32_t e, uint32_t m, int nb, uint32_t thr, uint32_t pkl, uint32_t one, uint32_t lane2) {
    const int ng = (nb + 7) >> 3;
    if (ng <= 0) return;
    uint32_t evA[8], evB[8];
    issue8x(e, m, 0, lane2, evA);
    int g = 0;
    while (true) {
        if (g + 1 < ng) issue8x(e, m, (g + 1) * 8, lane2, evB);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc.add(rl(m, g * 8 + u), evA[u], thr, pkl, one);      // the meta word straight from the lane into an SGPR
        if (++g >= ng) break;
        if (g + 1 < ng) issue8x(e, m, (g + 1) * 8, lane2, evA);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc.add(rl(m, g * 8 + u), evB[u], thr, pkl, one);
        if (++g >= ng) break;
    }
}

constexpr int WIX_WAVES = 4;
__global__ __launch_bounds__(WIX_WAVES * 64) void k_wave_ix(CountArgs a) {
    __shared__ __attribute__((aligned(4096))) uint32_t pk_all[WIX_WAVES][2 * 8 * 64];
    __shared__ WaveBook books[WIX_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* pk = pk_all[wv];
    WaveBook& book = books[wv];
    book_init(book, lane);
    for (int i = lane; i < 2 * 8 * 64; i += 64) pk[i] = 0;
    const uint32_t n_chunks = (uint32_t)a.scalars[SC_NCHUNK];
    const uint32_t thr = bq_threshold(a), pkl = lds_addr(pk + lane), lane2 = 2u * (uint32_t)lane;
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));                  // a register, not an inline constant (the asm block names it as an LDS data operand)
    unsigned long long nev_total = 0;
    const uint32_t n_waves_all = gridDim.x * WIX_WAVES;
    // a wave's first chunk is its own index, later ones come off the queue, several per dequeue (about 8 dequeues per wave)
    uint32_t qb = n_chunks / (n_waves_all * 8u);
    qb = qb < 1u ? 1u : (qb > 16u ? 16u : qb);
    uint32_t ck = blockIdx.x * WIX_WAVES + wv, ck_end = ck + 1;
    for (;; ++ck) {
        if (ck >= ck_end) {
            if (lane == 0) ck = (uint32_t)atomicAdd(&a.scalars[SC_QSMALL], (unsigned long long)qb) + n_waves_all;
            ck = rl(ck, 0); ck_end = ck + qb;
        }
        if (ck >= n_chunks) break;
        const uint32_t q0 = a.chunk_start[ck];
        const int nq = (int)(a.chunk_start[ck + 1] - q0);
        uint32_t s_w = 0, s_off = 0, s_cnt = 0; int2 s_geom = make_int2(0, 0);
        if (lane < nq) {
            const uint32_t s = a.slot_list[q0 + lane];
            s_w = a.slot_w[s]; s_off = a.slot_off[s]; s_cnt = a.slot_cnt[s];
            s_geom = a.ne_geom[s_w];
        }
        // the first 64 records of a slot are fetched one slot ahead
        uint2 cur = make_uint2(a.zero_lo, a.zero_hi);
        { const int n0 = (int)rl(s_cnt, 0); if (lane < n0) cur = a.rec[rl(s_off, 0) + lane]; }
        for (int qi = 0; qi < nq; ++qi) {
            const uint32_t w = rl(s_w, qi), src = rl(s_off, qi);
            const int n = (int)rl(s_cnt, qi);
            const int32_t tstart = (int32_t)rl((uint32_t)s_geom.x, qi);
            const uint32_t g = rl((uint32_t)s_geom.y, qi);
            const int tid = (int)(g & 0xffffffu), ct = (int)(g >> 24);
            uint2 nxt = make_uint2(a.zero_lo, a.zero_hi);
            if (qi + 1 < nq) { const int nn = (int)rl(s_cnt, qi + 1); if (lane < nn) nxt = a.rec[rl(s_off, qi + 1) + lane]; }
            int refb = 'N';
            { const int64_t pos = (int64_t)tstart + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            IxAcc acc; acc.init();
            for (int jb = 0; jb < n; jb += 64) {
                const int nb = n - jb < 64 ? n - jb : 64;
                uint2 r = cur;
                if (jb > 0) { r = make_uint2(a.zero_lo, a.zero_hi); if (lane < nb) r = a.rec[src + jb + lane]; }
                if (lane < nb) nev_total += meta_events(r.y);
                walk_regs_ix(acc, r.x, r.y, nb, thr, pkl, one, lane2);
            }
            acc.finish();
            uint32_t dp = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dp += pk[512 + k * 64 + lane] & 0xffffu;
            const IxCounters tot{pk, lane, dp - acc.nc};
            emit_unit<IxCounters, true>(a, tot, w, ct, tid, tstart, lane, &book, false, refb);
#pragma unroll
            for (int k = 0; k < 16; ++k) pk[k * 64 + lane] = 0;
            cur = nxt;
        }
    }
    for (int o = 32; o > 0; o >>= 1) nev_total += __shfl_down(nev_total, o);
    if (lane == 0) book.nev = nev_total;
    lds_fence();
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned long long nev = 0; uint32_t rt = 0, cols = 0, rdeep = 0, rsrc = 0;
        for (int w = 0; w < WIX_WAVES; ++w) {
            const WaveBook& b = books[w];
            if (lane < a.n_ct) rt += b.rows_true[lane];
            cols += b.cols; rdeep += b.rows_deep; rsrc += b.rows_src; nev += b.nev;
        }
        if (lane < a.n_ct && rt) atomicAdd(&a.scalars[SC_ROWS + lane], (unsigned long long)rt);
        if (lane == 0) {
            if (cols) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)cols);
            if (rdeep) atomicAdd(&a.scalars[SC_ROWS_DEEP], (unsigned long long)rdeep);
            if (rsrc) atomicAdd(&a.scalars[SC_ROWS_SRC + 0], (unsigned long long)rsrc);
            if (nev) { atomicAdd(&a.scalars[SC_EV_WAVE], nev); atomicAdd(&a.scalars[SC_EV_SRC + 0], nev); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Block path for slots with more than CAPW entries (or belonging to a multi-slot unit), in two
// kernels so that the event walk runs at full occupancy:
//   k_group_block  groups a slot's entries by barcode IN PLACE in global memory (LDS hash + scan)
//                  and records NSLICE run-aligned slice boundaries;
//   k_walk_block   4 waves walk one slice each straight from the grouped global arrays (coalesced
//                  record batches, 8 event loads in flight per wave), reduce in LDS, then emit the
//                  unit's rows or add into the multi-slot unit's global accumulators.
// Slots left with more than CAPB entries by a skewed barcode range go to k_pileup_huge.
struct GroupLds {
    uint32_t tkey[HB], tcnt[HB];
    uint32_t gcb[CAPB];
    uint32_t wave_tot[BLOCK_WAVES];
    uint32_t slot;
    unsigned long long nev;
};

__global__ __launch_bounds__(BLOCK_THREADS) void k_group_block(CountArgs a) {
    __shared__ GroupLds L;
    const int t = threadIdx.x;
    const uint32_t n_big = a.n_slots - (uint32_t)a.scalars[SC_NSMALL];
    unsigned long long nev = 0;
    for (uint32_t qi = blockIdx.x; qi < n_big; qi += gridDim.x) {     // static striding: slots are bounded, a shared queue word would cap the rate
        __syncthreads();
        const uint32_t s = a.slot_list[a.n_slots - 1 - qi];      // rejected items sit reversed at the end
        const int n = (int)a.slot_cnt[s];
        if (a.presorted && a.ne_nslot[a.slot_w[s]] > 1) continue;      // k_sort_deep wrote its records, slices and statistics
        if (n > CAPB) {
            if (t == 0) a.huge_list[atomicAdd(&a.scalars[SC_NHUGE], 1ull)] = s;
            continue;
        }
        const uint32_t src = a.slot_off[s];
        for (int i = t; i < n; i += BLOCK_THREADS) nev += meta_events(a.ent[src + i].x);      // events k_walk_block will read (statistics)
        group_by_cb<true, HB, CAPB, true>(a, src, n, L.gcb, nullptr, nullptr, L.tkey, L.tcnt, t, L.wave_tot);
        if (t <= NSLICE) {
            int j = (int)((int64_t)n * t / NSLICE);
            while (j > 0 && j < n && L.gcb[j] == L.gcb[j - 1]) ++j;
            a.slices[(uint64_t)s * (NSLICE + 1) + t] = (uint32_t)j;
        }
    }
    for (int o = 32; o > 0; o >>= 1) nev += __shfl_down(nev, o);
    __syncthreads();
    if (t == 0) L.nev = 0;
    __syncthreads();
    if ((t & 63) == 0 && nev) atomicAdd(&L.nev, nev);
    __syncthreads();
    if (t == 0 && L.nev) { atomicAdd(&a.scalars[SC_EV_DEEP], L.nev); atomicAdd(&a.scalars[SC_EV_SRC + 1], L.nev); }
}

// Walk grouped records [j0, j1) of the slot whose records start at src (all three wave-uniform).  The records are
// wave-uniform data: they are fetched eight at a time with SCALAR loads (constant address space: written by
// k_group_block, read-only here), so no vector register or readlane is spent on them, and the event loads of group
// g+1 are issued before group g is consumed (16 loads in flight per wave).
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define LSG_AS4 __attribute__((address_space(4)))
template <bool FULL>
__device__ __forceinline__ void issue8(const u32x16& R, int cnt, uint32_t lane2, uint32_t (&ev)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        ev[u] = 0;
        if (FULL || u < cnt) ev[u] = load_event(R[2 * u], R[2 * u + 1], lane2);      // (words past the slice are not records)
    }
}
template <bool FULL>
__device__ __forceinline__ void consume8(WalkAcc& acc, const u32x16& R, int cnt, const uint32_t (&ev)[8], uint32_t thr, uint32_t* pk, int lane) {
    acc.reserve(8, pk, lane);
    const uint32_t pkl = lds_addr(pk + lane);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (FULL || u < cnt) acc.add(R[2 * u + 1], ev[u], thr, pkl);
}
// the workgroup's shared planes (WalkLds): plane 0 quality sum [0..19] | forward count [20..31], plane 1 count [0..15] | duplicates
// [16..31]; good for slots of at most WALK_PLANE_MAX entries, never flushed, added to by all four waves at once
typedef PlaneAcc<20> WalkPlaneAcc;
constexpr int WALK_PLANE_MAX = 4095;
template <bool FULL>
__device__ __forceinline__ void consume8(WalkPlaneAcc& acc, const u32x16& R, int cnt, const uint32_t (&ev)[8], uint32_t thr, uint32_t* pk, int lane) {
    const uint32_t pkl = lds_addr(pk + lane);
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (FULL || u < cnt) acc.add(R[2 * u + 1], ev[u], thr, pkl, one);
}
template <class ACC>
__device__ __forceinline__ void walk_global(const CountArgs& a, ACC& acc, uint32_t src, int j0, int j1, uint32_t* pk, int lane) {
    const uint32_t thr = bq_threshold(a), lane2 = 2u * (uint32_t)lane;
    const int n = j1 - j0;
    if (n <= 0) return;
    const LSG_AS4 u32x16* p = (const LSG_AS4 u32x16*)(uintptr_t)(a.rec + src + j0);
    const int ngf = n >> 3, rem = n & 7;                              // full groups, entries of the partial last group
    if (ngf > 0) {
        u32x16 RA = p[0], RB;
        uint32_t evA[8], evB[8];
        issue8<true>(RA, 8, lane2, evA);
        int g = 0;
        while (true) {
            if (g + 1 < ngf) { RB = p[g + 1]; issue8<true>(RB, 8, lane2, evB); }      // A = group g, its loads in flight
            consume8<true>(acc, RA, 8, evA, thr, pk, lane);
            if (++g >= ngf) break;
            if (g + 1 < ngf) { RA = p[g + 1]; issue8<true>(RA, 8, lane2, evA); }      // B = group g
            consume8<true>(acc, RB, 8, evB, thr, pk, lane);
            if (++g >= ngf) break;
        }
    }
    if (rem) {
        const u32x16 R = p[ngf];
        uint32_t ev[8];
        issue8<false>(R, rem, lane2, ev);
        consume8<false>(acc, R, rem, ev, thr, pk, lane);
    }
}

constexpr int WALK_THREADS = NSLICE * 64;
struct WalkLdsW { uint32_t pk[NSLICE][8 * 64]; uint32_t acc[NCTR][64]; };      // slots of more than WALK_PLANE_MAX entries: per-wave packed counters, flushed
struct WalkLdsP { uint32_t plane[2][8 * 64]; uint32_t nc[64]; };                  // everything else: two shared planes
template <bool PLANES> struct WalkLds;
template <> struct alignas(4096) WalkLds<true> { WalkLdsP p; uint32_t slot; WaveBook book; };
template <> struct alignas(4096) WalkLds<false> { WalkLdsW w; uint32_t slot; WaveBook book; };
// a slot's finished counters read from the shared planes
struct PlaneCounters {
    const uint32_t* pl; const uint32_t* ncw; int lane;
    __device__ __forceinline__ uint32_t BC(int k) const { return pl[512 + k * 64 + lane] & 0xffffu; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return pl[512 + k * 64 + lane] >> 16; }
    __device__ __forceinline__ uint32_t BQ(int k) const { return pl[k * 64 + lane] & 0xfffffu; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return pl[k * 64 + lane] >> 20; }
    __device__ __forceinline__ uint32_t NCDUP() const {
        uint32_t dp = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) dp += pl[512 + k * 64 + lane] & 0xffffu;
        return dp - ncw[lane];
    }
};

// PLANES: every slot of the big list with at most WALK_PLANE_MAX entries; the others are handed to the second launch (!PLANES: the
// flushing per-wave counters, any size) through rest_list.
template <bool PLANES>
__global__ __launch_bounds__(WALK_THREADS) __attribute__((amdgpu_num_sgpr(96))) void k_walk_block(CountArgs a) {
    __shared__ WalkLds<PLANES> L;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (wv == 0) { book_init(L.book, lane); if (lane == 0) L.book.src = 1; }
    const uint32_t n_big = PLANES ? a.n_slots - (uint32_t)a.scalars[SC_NSMALL] : (uint32_t)a.scalars[SC_NREST];
    for (bool first = true;; first = false) {
        __syncthreads();
        if (t == 0) L.slot = first ? blockIdx.x : (uint32_t)atomicAdd(&a.scalars[PLANES ? SC_QBIG : SC_QREST], 1ull) + gridDim.x;   // first item = own index
        __syncthreads();
        const uint32_t qi = rl(L.slot, 0);
        if (qi >= n_big) break;
        const uint32_t s = rl(PLANES ? a.slot_list[a.n_slots - 1 - qi] : a.rest_list[qi], 0);
        const int n = (int)rl(a.slot_cnt[s], 0);
        const uint32_t w = rl(a.slot_w[s], 0), src = rl(a.slot_off[s], 0);
        if (n > CAPB && !(a.presorted && a.ne_nslot[w] > 1)) continue;   // k_pileup_huge's slot
        if (PLANES && n > WALK_PLANE_MAX) {
            if (t == 0) a.rest_list[atomicAdd(&a.scalars[SC_NREST], 1ull)] = s;
            continue;
        }
        const int j0 = (int)rl(a.slices[(uint64_t)s * (NSLICE + 1) + wv], 0), j1 = (int)rl(a.slices[(uint64_t)s * (NSLICE + 1) + wv + 1], 0);
        const bool multi = a.ne_nslot[w] > 1;
        if constexpr (PLANES) {
            // the four waves add straight into the workgroup's two planes: no per-wave counters, no flush, no merge
            for (int i = t; i < 2 * 8 * 64 + 64; i += WALK_THREADS) (&L.p.plane[0][0])[i] = 0;
            __syncthreads();
            WalkPlaneAcc acc; acc.init();
            walk_global(a, acc, src, j0, j1, &L.p.plane[0][0], lane);
            acc.finish();
            if (acc.nc) atomicAdd(&L.p.nc[lane], acc.nc);
            __syncthreads();
            const PlaneCounters tot{&L.p.plane[0][0], L.p.nc, lane};
            if (multi) {
                // multi-slot unit: this slot's partial sums go to its own slab (the accumulator rows k_finalize_multi adds up;
                // global atomics here cost more than the whole walk)
                uint32_t* dst = a.macc + (uint64_t)(a.ne_acc[w] + (s - a.ne_slot_base[w])) * (NCTR * 64);
                const uint32_t* pl = &L.p.plane[0][0];
                if (wv == 0) dst[lane] = tot.NCDUP();
                for (int r = 1 + wv; r < NCTR; r += NSLICE) {
                    const int k = (r - 1) & 7, grp = (r - 1) >> 3;                      // rows 1..8 dup, 9..16 count, 17..24 quality, 25..32 forward
                    const uint32_t lo = pl[k * 64 + lane], hi = pl[512 + k * 64 + lane];
                    dst[r * 64 + lane] = grp == 0 ? hi >> 16 : grp == 1 ? hi & 0xffffu : grp == 2 ? lo & 0xfffffu : lo >> 20;
                }
            } else if (wv == 0) {
                const int2 geom = a.ne_geom[w];
                emit_unit(a, tot, w, (int)((uint32_t)geom.y >> 24), geom.y & 0xffffff, geom.x, lane, &L.book, true);
            }
        } else {
            uint32_t* pk = L.w.pk[wv];
            for (int i = lane; i < 8 * 64; i += 64) pk[i] = 0;
            for (int i = t; i < NCTR * 64; i += WALK_THREADS) (&L.w.acc[0][0])[i] = 0;
            __syncthreads();
            WalkAcc acc; acc.init(&L.w.acc[0][0]);
            walk_global(a, acc, src, j0, j1, pk, lane);
            acc.finish(pk, lane);
            __syncthreads();
            if (multi) {
                uint32_t* dst = a.macc + (uint64_t)(a.ne_acc[w] + (s - a.ne_slot_base[w])) * (NCTR * 64);
                for (int i = t; i < NCTR * 64; i += WALK_THREADS) dst[i] = (&L.w.acc[0][0])[i];
            } else if (wv == 0) {
                const LdsCounters tot{&L.w.acc[0][0], lane};
                const int2 geom = a.ne_geom[w];
                emit_unit(a, tot, w, (int)((uint32_t)geom.y >> 24), geom.y & 0xffffff, geom.x, lane, &L.book, true);
            }
        }
    }
    if (wv == 0) book_flush(a, L.book, lane);                      // the events this kernel reads are counted by k_group_block
}

// ------------------------------------------------------------------------------------------------
// Huge-slot kernel: one 512-thread workgroup per slot with > CAPW entries or belonging to a multi-slot
// unit.  Normal case (<= CAPB entries): one staged pass.  Fallback (a skewed barcode range left
// more than CAPB entries in a slot): passes over coarse barcode buckets; a bucket that alone
// exceeds CAPB is streamed barcode by barcode by wave 0.
struct alignas(2048) BlockLds {
    uint32_t gkey[CAPB], gev[CAPB], gmeta[CAPB];
    union {
        struct { uint32_t tkey[HB], tcnt[HB]; } h;
        struct { uint32_t pk[BLOCK_WAVES][8 * 64]; uint32_t acc[NCTR][64]; } w;
    } u;
    uint32_t hist[NBUCKET];
    uint32_t pass_lo[NBUCKET + 1];
    uint32_t wave_tot[BLOCK_WAVES];
    uint32_t n_pass, scount, slot;
    WaveBook book;
};
static_assert(sizeof(uint32_t) * (BLOCK_WAVES * 8 * 64 + NCTR * 64) <= sizeof(uint32_t) * 2 * HB, "pk+acc must fit in the hash arrays");

// stage + group the entries of [src, src+n) whose bucket lies in [b_lo, b_hi) (filter) or all of them
__device__ __forceinline__ int block_stage_filtered(const CountArgs& a, BlockLds& L, uint32_t src, int n, int shift, uint32_t b_lo, uint32_t b_hi, int t) {
    // compact matching entries into gkey/gev/gmeta (unordered), then regroup in place via a second
    // buffer-free trick: entries are re-read from LDS into registers by group_by_cb-style code below.
    __syncthreads();
    if (t == 0) L.scount = 0;
    __syncthreads();
    for (int i = t; i < n; i += BLOCK_THREADS) {
        const uint4 v = unpack_entry(a, a.ent[src + i]);
        uint32_t b = (v.x & CB_MASK) >> shift;
        if (b >= b_lo && b < b_hi) {
            uint32_t slot = atomicAdd(&L.scount, 1u);
            L.gkey[slot] = v.x; L.gev[slot] = v.y; L.gmeta[slot] = v.z;
        }
    }
    __syncthreads();
    const int ns = (int)L.scount;
    // regroup the staged entries by barcode (registers hold the entries while the hash is built)
    constexpr int RMAX = CAPB / BLOCK_THREADS;
    uint32_t ek[RMAX], ee[RMAX], em[RMAX], hs[RMAX];
    for (int i = t; i < HB; i += BLOCK_THREADS) { L.u.h.tkey[i] = KEY_INVALID; L.u.h.tcnt[i] = 0; }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        int i = t + r * BLOCK_THREADS;
        ek[r] = KEY_INVALID; ee[r] = 0; em[r] = 0; hs[r] = 0;
        if (i < ns) { ek[r] = L.gkey[i]; ee[r] = L.gev[i]; em[r] = L.gmeta[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (t + r * BLOCK_THREADS < ns) {
            uint32_t cb = ek[r] & CB_MASK;
            uint32_t h = hash_cb(cb) >> (32 - __builtin_ctz(HB));
            while (true) {
                uint32_t prev = atomicCAS(&L.u.h.tkey[h], KEY_INVALID, cb);
                if (prev == KEY_INVALID || prev == cb) break;
                h = (h + 1) & (HB - 1);
            }
            hs[r] = h | (atomicAdd(&L.u.h.tcnt[h], 1u) << 16);
        }
    }
    __syncthreads();
    constexpr int PER = HB / BLOCK_THREADS;
    uint32_t loc[PER]; uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = L.u.h.tcnt[t * PER + q]; sum += loc[q]; }
    uint32_t incl = sum; const int lane = t & 63, wv = t >> 6;
    for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    uint32_t excl = incl - sum;
    if (lane == 63) L.wave_tot[wv] = incl;
    __syncthreads();
    for (int q = 0; q < wv; ++q) excl += L.wave_tot[q];
#pragma unroll
    for (int q = 0; q < PER; ++q) { L.u.h.tcnt[t * PER + q] = excl; excl += loc[q]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (t + r * BLOCK_THREADS < ns) {
            uint32_t p = L.u.h.tcnt[hs[r] & 0xffffu] + (hs[r] >> 16);
            L.gkey[p] = ek[r]; L.gev[p] = ee[r]; L.gmeta[p] = em[r];
        }
    }
    __syncthreads();
    for (int q = t; q < ns; q += BLOCK_THREADS)                       // first entry of every barcode run
        if (q == 0 || (L.gkey[q - 1] & CB_MASK) != (L.gkey[q] & CB_MASK)) atomicOr(&L.gmeta[q], META_NEWRUN);
    __syncthreads();
    return ns;
}

__device__ __forceinline__ void block_walk_slices(const CountArgs& a, BlockLds& L, Acc& acc, int ns, int wv, int lane) {
    // run-aligned slice of this wave
    int j0 = (int)((int64_t)ns * wv / BLOCK_WAVES), j1 = (int)((int64_t)ns * (wv + 1) / BLOCK_WAVES);
    while (j0 > 0 && j0 < ns && (L.gkey[j0] & CB_MASK) == (L.gkey[j0 - 1] & CB_MASK)) ++j0;
    while (j1 > 0 && j1 < ns && (L.gkey[j1] & CB_MASK) == (L.gkey[j1 - 1] & CB_MASK)) ++j1;
    if (j0 > j1) j0 = j1;
    j0 = (int)rl((uint32_t)j0, 0); j1 = (int)rl((uint32_t)j1, 0);
    // the hash arrays are dead now: pk lives there
    __syncthreads();
    uint32_t* pk = L.u.w.pk[wv];
    for (int i = lane; i < 8 * 64; i += 64) pk[i] = 0;
    acc.new_run();
    walk(a, acc, L.gev, L.gmeta, j0, j1, pk, lane);
    acc.finish(pk, lane);
    acc.new_run();
}

__global__ __launch_bounds__(BLOCK_THREADS) void k_pileup_huge(CountArgs a) {
    __shared__ BlockLds L;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (wv == 0) { book_init(L.book, lane); if (lane == 0) L.book.src = 2; }
    const uint32_t n_huge = (uint32_t)a.scalars[SC_NHUGE];
    int shift = 0;
    while (((uint32_t)(a.n_cb - 1) >> shift) >= (uint32_t)NBUCKET) ++shift;
    unsigned long long nev_total = 0;

    while (true) {
        __syncthreads();
        if (t == 0) L.slot = (uint32_t)atomicAdd(&a.scalars[SC_QHUGE], 1ull);
        __syncthreads();
        const uint32_t qi = L.slot;
        if (qi >= n_huge) break;
        const uint32_t s = a.huge_list[qi];
        const uint32_t w = a.slot_w[s], src = a.slot_off[s];
        const int n = (int)a.slot_cnt[s];
        const int2 geom = a.ne_geom[w];
        const int32_t tstart = geom.x; const int tid = geom.y & 0xffffff, ct = (int)((uint32_t)geom.y >> 24);
        const bool multi = a.ne_nslot[w] > 1;
        Acc acc; acc.init();

        if (n <= CAPB) {
            group_by_cb<true, HB, CAPB>(a, src, n, L.gkey, L.gev, L.gmeta, L.u.h.tkey, L.u.h.tcnt, t, L.wave_tot);
            block_walk_slices(a, L, acc, n, wv, lane);
        } else {
            // fallback passes over coarse barcode buckets
            __syncthreads();
            for (int i = t; i < NBUCKET; i += BLOCK_THREADS) L.hist[i] = 0;
            __syncthreads();
            for (int i = t; i < n; i += BLOCK_THREADS) atomicAdd(&L.hist[(a.ent[src + i].x & CB_MASK) >> shift], 1u);
            __syncthreads();
            if (t == 0) {
                uint32_t np = 0, cur = 0; bool open = false;
                for (uint32_t b = 0; b < (uint32_t)NBUCKET; ++b) {
                    uint32_t h = L.hist[b];
                    if (h == 0) continue;
                    if (!open || cur + h > (uint32_t)CAPB || h > (uint32_t)CAPB) { L.pass_lo[np++] = b; cur = 0; open = true; }
                    cur += h;
                    if (h > (uint32_t)CAPB) open = false;
                }
                L.pass_lo[np] = NBUCKET;
                L.n_pass = np;
            }
            __syncthreads();
            const uint32_t n_pass = L.n_pass;
            for (uint32_t p = 0; p < n_pass; ++p) {
                const uint32_t b_lo = L.pass_lo[p], b_hi = L.pass_lo[p + 1];
                if (L.hist[b_lo] <= (uint32_t)CAPB) {
                    const int ns = block_stage_filtered(a, L, src, n, shift, b_lo, b_hi, t);
                    block_walk_slices(a, L, acc, ns, wv, lane);
                } else {
                    // stream mode: wave 0 walks the bucket barcode by barcode straight from global memory
                    __syncthreads();
                    if (wv == 0) {
                        uint32_t* pk = L.u.w.pk[0];
                        for (int i = lane; i < 8 * 64; i += 64) pk[i] = 0;
                        const uint32_t c_lo = b_lo << shift, c_hi = (b_lo + 1) << shift;
                        for (uint32_t c = c_lo; c < c_hi && c < (uint32_t)a.n_cb; ++c) {
                            acc.new_run();
                            for (int ib = 0; ib < n; ib += 64) {
                                uint32_t k = KEY_INVALID, e = 0, m = 0;
                                if (ib + lane < n) { const uint4 v = unpack_entry(a, a.ent[src + ib + lane]); k = v.x; e = v.y; m = v.z; }
                                bool match = k != KEY_INVALID && (k & CB_MASK) == c;
                                if (match) acc.nev += meta_events(m);
                                unsigned long long mm = __ballot(match);
                                while (mm) {
                                    int l = __ffsll((long long)mm) - 1; mm &= mm - 1;
                                    const uint32_t ms = rl(m, l);                        // no META_NEWRUN here: the run is the whole barcode, reset by new_run() above
                                    const uint32_t evv = load_event(rl(e, l), ms, 2u * (uint32_t)lane);
                                    acc.reserve(1, pk, lane);
                                    acc.add(ms, evv, bq_threshold(a), lds_addr(pk + lane));
                                }
                            }
                        }
                        acc.finish(pk, lane);
                        acc.new_run();
                    }
                    __syncthreads();
                }
            }
        }
        nev_total += acc.nev;
        // reduce the waves' accumulators (hash arrays are dead; acc aliases them next to pk)
        __syncthreads();
        for (int i = t; i < NCTR * 64; i += BLOCK_THREADS) (&L.u.w.acc[0][0])[i] = 0;
        __syncthreads();
        atomicAdd(&L.u.w.acc[0][lane], acc.ncdup);
#pragma unroll
        for (int sy = 0; sy < 8; ++sy) {
            atomicAdd(&L.u.w.acc[1 + sy][lane], acc.dup[sy]);
            atomicAdd(&L.u.w.acc[9 + sy][lane], acc.bc[sy]);
            atomicAdd(&L.u.w.acc[17 + sy][lane], acc.bq[sy]);
            atomicAdd(&L.u.w.acc[25 + sy][lane], acc.bcf[sy]);
        }
        __syncthreads();
        if (multi) {
            uint32_t* dst = a.macc + (uint64_t)(a.ne_acc[w] + (s - a.ne_slot_base[w])) * (NCTR * 64);
            for (int i = t; i < NCTR * 64; i += BLOCK_THREADS) dst[i] = (&L.u.w.acc[0][0])[i];
        } else if (wv == 0) {
            Acc tot; tot.init();
            tot.ncdup = L.u.w.acc[0][lane];
#pragma unroll
            for (int sy = 0; sy < 8; ++sy) {
                tot.dup[sy] = L.u.w.acc[1 + sy][lane]; tot.bc[sy] = L.u.w.acc[9 + sy][lane];
                tot.bq[sy] = L.u.w.acc[17 + sy][lane]; tot.bcf[sy] = L.u.w.acc[25 + sy][lane];
            }
            emit_unit(a, tot, w, ct, tid, tstart, lane, &L.book, true);
        }
    }
    if (wv == 0) book_flush(a, L.book, lane);
    for (int o = 32; o > 0; o >>= 1) nev_total += __shfl_down(nev_total, o);
    if (lane == 0 && nev_total) { atomicAdd(&a.scalars[SC_EV_DEEP], nev_total); atomicAdd(&a.scalars[SC_EV_SRC + 2], nev_total); }
}

// multi-slot units: sum the unit's partial-sum slabs (8 waves, each a stride of the slots), gates + emission by wave 0
constexpr int FIN_THREADS = 512;
__global__ __launch_bounds__(FIN_THREADS) void k_finalize_multi(CountArgs a) {
    __shared__ uint32_t sacc[NCTR][64];
    __shared__ WaveBook book;             // rows from a workgroup arena, exact counters flushed once (not five global atomics per unit)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (wv == 0) { book_init(book, lane); if (lane == 0) book.src = 3; }
    for (uint32_t k = blockIdx.x; k < a.n_multi; k += gridDim.x) {
        const uint32_t w = a.multi_list[k];
        const uint32_t nslot = a.ne_nslot[w];
        { const uint32_t tile = a.ne_units[w] / (uint32_t)a.n_ct; if (tile < a.tile_lo || tile >= a.tile_hi) continue; }   // (the tile-major lists are static: all tiles)
        __syncthreads();
        for (int i = t; i < NCTR * 64; i += FIN_THREADS) (&sacc[0][0])[i] = 0;
        __syncthreads();
        Acc tot; tot.init();
        for (uint32_t j = wv; j < nslot; j += FIN_THREADS / 64) {      // the unit's slabs are contiguous
            const uint32_t* src = a.macc + (uint64_t)(a.ne_acc[w] + j) * (NCTR * 64);
            tot.ncdup += src[lane];
#pragma unroll
            for (int sy = 0; sy < 8; ++sy) {
                tot.dup[sy] += src[(1 + sy) * 64 + lane]; tot.bc[sy] += src[(9 + sy) * 64 + lane];
                tot.bq[sy] += src[(17 + sy) * 64 + lane]; tot.bcf[sy] += src[(25 + sy) * 64 + lane];
            }
        }
        if ((uint32_t)wv < nslot) {
            atomicAdd(&sacc[0][lane], tot.ncdup);
#pragma unroll
            for (int sy = 0; sy < 8; ++sy) {
                atomicAdd(&sacc[1 + sy][lane], tot.dup[sy]); atomicAdd(&sacc[9 + sy][lane], tot.bc[sy]);
                atomicAdd(&sacc[17 + sy][lane], tot.bq[sy]); atomicAdd(&sacc[25 + sy][lane], tot.bcf[sy]);
            }
        }
        __syncthreads();
        if (wv == 0) {
            tot.ncdup = sacc[0][lane];
#pragma unroll
            for (int sy = 0; sy < 8; ++sy) {
                tot.dup[sy] = sacc[1 + sy][lane]; tot.bc[sy] = sacc[9 + sy][lane];
                tot.bq[sy] = sacc[17 + sy][lane]; tot.bcf[sy] = sacc[25 + sy][lane];
            }
            const int2 geom = a.ne_geom[w];
            emit_unit(a, tot, w, (int)((uint32_t)geom.y >> 24), geom.y & 0xffffff, geom.x, lane, &book, true);
        }
    }
    if (wv == 0) book_flush(a, book, lane);
}

// Work-balanced chunks of the wave kernel's slot list.  work(slot) = entries + WORK_W0; chunk k holds the
// slots whose exclusive work prefix lies in [k*E, (k+1)*E), E chosen so that every wave gets several chunks.
constexpr uint32_t WORK_W0 = 16;
constexpr uint32_t CHUNK_EMIN = CAPW + WORK_W0, CHUNK_EMAX = (uint32_t)QCHUNK * (WORK_W0 + 1) - 1;
struct SlotWork {
    const uint32_t* slot_list; const uint32_t* slot_cnt; const unsigned long long* scalars;
    __host__ __device__ uint32_t operator()(const uint32_t& i) const {
        return i < (uint32_t)scalars[SC_NSMALL] ? slot_cnt[slot_list[i]] + WORK_W0 : 0u;
    }
};
__global__ void k_chunk_starts(CountArgs a, uint32_t n_waves) {
    const uint32_t n_small = (uint32_t)a.scalars[SC_NSMALL];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_small) return;
    uint32_t E = a.slot_pex[a.n_slots] / (n_waves * 4u);
    E = E < CHUNK_EMIN ? CHUNK_EMIN : (E > CHUNK_EMAX ? CHUNK_EMAX : E);
    const uint32_t c = a.slot_pex[i] / E;
    const int64_t cp = i ? (int64_t)(a.slot_pex[i - 1] / E) : -1;
    for (int64_t k = cp + 1; k <= (int64_t)c; ++k) a.chunk_start[k] = i;
    if (i == n_small - 1) { a.chunk_start[c + 1] = n_small; a.scalars[SC_NCHUNK] = (unsigned long long)c + 1; }
}

// ------------------------------------------------------------------------------------------------
struct NonEmpty {
    const uint32_t* cnt;
    __host__ __device__ bool operator()(const uint32_t& i) const { return cnt[i] != 0; }
};
struct SmallSlot {   // slot handled by the wave kernel
    const uint32_t* slot_cnt; const uint32_t* slot_w; const uint32_t* ne_nslot;
    __host__ __device__ bool operator()(const uint32_t& s) const { return slot_cnt[s] <= (uint32_t)CAPW && ne_nslot[slot_w[s]] == 1; }
};
struct MultiUnit {
    const uint32_t* ne_nslot;
    __host__ __device__ bool operator()(const uint32_t& w) const { return ne_nslot[w] > 1; }
};

static int read_scalars(lsg_ctx* c, unsigned long long* sc) {
    // pinned landing zone: a pageable destination costs a staging copy and tens of microseconds per read
    LSG_HIP(hipMemcpyAsync(c->h_pin, c->d_scalars.p, SC_COUNT * 8, hipMemcpyDeviceToHost, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    memcpy(sc, c->h_pin, SC_COUNT * 8);
    return 0;
}

static void fill_args(lsg_ctx* c, const lsg_count_params* p, CountArgs& a) {
    a.n_reads = c->rd.n_reads; a.n_segs = c->rd.n_segs;
    a.read_tid = c->rd.read_tid; a.read_flag = c->rd.read_flag; a.read_mapq = c->rd.read_mapq; a.read_cb = c->rd.read_cb;
    a.seg_read = c->rd.seg_read; a.seg_start = c->rd.seg_start; a.seg_len = c->rd.seg_len; a.seg_ev_off = c->rd.seg_ev_off;
    a.events = c->rd.events;
    {   // layout.hip leaves >= 256 zero bytes behind the last tile slot: a whole line of "no event here"
        const uint64_t z = (uint64_t)(uintptr_t)(c->rd.events + c->rd.n_events);
        a.zero_lo = (uint32_t)z; a.zero_hi = (uint32_t)(z >> 32) & 0x7fffu;
    }
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.ref_ptr = c->d_ref_ptrs.as<const uint8_t*>(); a.celltype_of = c->d_celltype_of.as<uint8_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.n_ct = c->n_ct;
    a.n_units = c->n_tiles * (uint32_t)c->n_ct;
    a.tile_lo = c->tile_lo; a.tile_hi = c->tile_hi;
    a.min_bq = p->min_bq; a.min_mq = p->min_mq; a.min_dp = p->min_dp; a.min_cc = p->min_cc;
    a.ignore_orphans = p->ignore_orphans; a.flag_exclude = p->flag_exclude;
    a.read_key = c->d_read_key.as<uint32_t>(); a.unit_cnt = c->d_unit_cnt.as<uint32_t>();
    a.unit_off = c->d_unit_off.as<uint32_t>(); a.unit_cursor = c->d_unit_fill.as<uint32_t>(); a.ent_half = c->entries_upper + 1;
    a.ct_rank = c->d_ct_rank.as<uint32_t>();
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) a.ct_size[i] = c->ct_size[i] ? c->ct_size[i] : 1u;
    a.ne_units = c->d_ne_units.as<uint32_t>(); a.ne_nslot = c->ws[WS_NE_NSLOT].as<uint32_t>();
    a.ne_slot_base = c->ws[WS_NE_SLOT_BASE].as<uint32_t>(); a.ne_acc = c->ws[WS_NE_ACC].as<uint32_t>();
    a.ne_geom = c->ws[WS_NE_GEOM].as<int2>();
    a.ne_mask = c->d_ne_mask.as<uint64_t>(); a.ne_rowbase = c->d_ne_rowbase.as<uint32_t>();
    a.slot_w = c->ws[WS_SLOT_W].as<uint32_t>(); a.slot_cnt = c->ws[WS_SLOT_CNT].as<uint32_t>();
    a.slot_off = c->ws[WS_SLOT_OFF].as<uint32_t>();
    a.ent = c->ws[WS_ENT].as<uint2>(); a.rec = c->ws[WS_REC].as<uint2>(); a.seg_info = c->ws[WS_SEG_INFO].as<uint2>();
    a.slot_list = c->ws[WS_SLOT_LIST].as<uint32_t>(); a.multi_list = c->ws[WS_MULTI_LIST].as<uint32_t>();
    a.macc = c->ws[WS_MACC].as<uint32_t>();
    a.slot_pex = c->ws[WS_SLOT_PEX].as<uint32_t>(); a.chunk_start = c->ws[WS_CHUNK_START].as<uint32_t>();
    a.slices = c->ws[WS_SLICES].as<uint32_t>(); a.huge_list = c->ws[WS_HUGE_LIST].as<uint32_t>();
    a.rest_list = a.huge_list + (c->entries_upper / CAPB + 16);      // second half: slots the plane walk hands on
    a.n_ne = c->n_ne; a.n_slots = c->n_slots; a.n_multi = c->n_multi;
    { uint32_t mx = 0; for (int i = 0; i < c->n_ct; ++i) mx = c->ct_size[i] > mx ? c->ct_size[i] : mx; a.presorted = mx <= (uint32_t)SORT_RMAX && !getenv("LSG_NO_PRESORT") ? 1u : 0u; }
    a.scalars = c->d_scalars.as<unsigned long long>();
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) a.rows[i] = c->d_rows[i].as<uint32_t>();
    a.row_cap = c->row_cap; a.arena = c->arena;
    a.two_ended = c->n_ct <= 2 && !getenv("LSG_COUNT_PASS") ? 1u : 0u;
    a.inline_seg_info = a.two_ended && !getenv("LSG_SEG_INFO_KERNEL") ? 1u : 0u;
    a.tile_off = c->d_tile_off.as<uint32_t>(); a.cur_lo = c->d_cur_lo.as<uint32_t>(); a.cur_hi = c->d_cur_hi.as<uint32_t>();
    a.read_drop = c->has_drops ? c->d_read_drop.as<uint8_t>() : nullptr;
    a.ix0 = c->d_ix0.as<uint32_t>(); a.ix1 = c->d_ix1.as<uint32_t>(); a.ix2 = c->d_ix2.as<uint32_t>();
    a.ix_netile = c->d_ix_netile.as<uint32_t>(); a.ix_chunk = c->d_ix_chunk.as<int32_t>(); a.ix_carry = c->d_ix_carry.as<unsigned long long>();
    a.ix_n = c->ix_n; a.index_path = c->index_path ? 1u : 0u;
    a.ixb_key = nullptr; a.ixb_read = nullptr;
    if (c->index_path) a.presorted = 1u;                  // the records arrive grouped by barcode: no k_sort_deep / k_group_block
}

// launch-shape knobs for tuning runs (environment overrides; the defaults are what ships)
static int tune_int(const char* name, int dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const int x = atoi(v);
    return x > 0 && x <= 64 ? x : dflt;
}

static int cub_tmp(lsg_ctx* c, size_t bytes) { return c->d_cub_tmp.reserve(bytes + 256); }

#define SCAN_U32(in, out, n)                                                                              \
    do {                                                                                                  \
        size_t tb_ = 0;                                                                                   \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_, (in), (out), (int)(n), st));              \
        if (cub_tmp(c, tb_)) return -1;                                                                   \
        tb_ = c->d_cub_tmp.cap;                                                                           \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb_, (in), (out), (int)(n), st));       \
    } while (0)

// entries a tile can ever hold + their prefix (static until the reads or the contigs change)
static int tile_capacities(lsg_ctx* c) {
    if (c->tile_caps_valid) return 0;
    hipStream_t st = c->stream;
    const size_t nt = (size_t)c->n_tiles + 2;
    const int64_t S = c->rd.n_segs;
    if (c->d_tile_cap.reserve(nt * 4) || c->d_tile_off.reserve(nt * 4) || c->d_cur_lo.reserve(nt * 4) || c->d_cur_hi.reserve(nt * 4) ||
        c->ws[WS_SEG_INFO].reserve(((size_t)S + 1) * 8) || c->d_scalars.reserve(SC_COUNT * 8)) return -1;
    LSG_HIP(hipMemsetAsync(c->d_tile_cap.p, 0, nt * 4, st));
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, st));
    if (S > 0) {
        CountArgs a{};
        a.n_reads = c->rd.n_reads; a.n_segs = S;
        a.read_tid = c->rd.read_tid; a.read_cb = c->rd.read_cb; a.read_flag = c->rd.read_flag;
        a.seg_read = c->rd.seg_read; a.seg_start = c->rd.seg_start; a.seg_len = c->rd.seg_len;
        a.tile_base = c->d_tile_base.as<uint32_t>(); a.contig_len = c->d_contig_len.as<int64_t>(); a.n_contigs = c->n_contigs;
        a.n_ct = 1; a.tile_lo = 0; a.tile_hi = c->n_tiles;
        a.seg_info = c->ws[WS_SEG_INFO].as<uint2>(); a.unit_cnt = c->d_tile_cap.as<uint32_t>();
        a.scalars = c->d_scalars.as<unsigned long long>();
        unsigned g = (unsigned)((S + 255) / 256); if (g > (unsigned)(c->n_cus * 16)) g = (unsigned)(c->n_cus * 16);
        hipLaunchKernelGGL(k_seg_info_static, dim3(g), dim3(256), 0, st, a);
        unsigned seg_grid = (unsigned)((S + 256 * BIN_SUPER - 1) / (256 * BIN_SUPER));
        if (seg_grid > (unsigned)(c->n_cus * 8)) seg_grid = (unsigned)(c->n_cus * 8);
        hipLaunchKernelGGL(k_bin_segments<0>, dim3(seg_grid), dim3(256), 0, st, a);
    }
    SCAN_U32(c->d_tile_cap.as<uint32_t>(), c->d_tile_off.as<uint32_t>(), c->n_tiles + 1);
    LSG_HIP(hipGetLastError());
    c->tile_caps_valid = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Tile index (static per load).  A tile's entries in barcode order do not depend on the count's parameters or on the barcode ->
// cell-type table, so they are sorted ONCE: the scatter of k_bin_segments<2> over the parameter-free admission record (every read
// with a barcode, one cell type) fills the tiles' regions in arrival order and writes a sort key (tile << 24 | barcode) and the
// owning read beside every entry; a radix sort of the keys gives the permutation; k_ix_gather writes the entries in that order
// together with what a count needs to decide admission on its own: the read's SAM flag bits and MAPQ.
constexpr int IX_CHUNK = 2048;            // static entries resolved by one workgroup of k_resolve
constexpr uint32_t IX_RUNSTART = 1u << 31, IX_TILESTART = 1u << 20, IX_SEGSTART = 1u << 21;

__global__ void k_ix_gather(const uint64_t* key, const uint32_t* perm, const uint2* ent, const uint32_t* eread, const uint16_t* read_flag,
                            const uint8_t* read_mapq, uint64_t n, uint32_t* ix0, uint32_t* ix1, uint32_t* ix2) {
    const uint64_t n_pad = (n + IX_CHUNK - 1) / IX_CHUNK * IX_CHUNK + 16;       // whole chunks + the look-ahead word: entries no count admits
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i >= n) { ix0[i] = CB_MASK | IX_RUNSTART; ix1[i] = 0; ix2[i] = IX_TILESTART; continue; }
        const uint32_t p = perm[i];
        const uint2 e = ent[p];
        const uint32_t rr = eread[p], r = rr & 0x7fffffffu;
        const uint64_t k = key[i], kp = i ? key[i - 1] : ~0ull;
        ix0[i] = (e.x & 0x7fffffffu) | (k != kp ? IX_RUNSTART : 0u);
        ix1[i] = e.y;
        ix2[i] = ((uint32_t)read_flag[r] & 0xfffu) | ((uint32_t)read_mapq[r] << 12) | ((k >> 24) != (kp >> 24) ? IX_TILESTART : 0u) | (rr >> 31 ? IX_SEGSTART : 0u);
    }
}
struct CapNonZero {
    const uint32_t* cap;
    __host__ __device__ bool operator()(const uint32_t& t) const { return cap[t] != 0; }
};
// chunk k starts at static entry k * IX_CHUNK: index (into the list of non-empty tiles) of the tile holding that entry, minus one when
// the entry is the tile's first, so that tile(entry) = netile[chunk[k] + tile starts among the chunk's entries up to and including it]
__global__ void k_ix_chunks(const uint32_t* netile, uint32_t n_netile, const uint32_t* tile_off, uint64_t n, int32_t* chunk) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = k * IX_CHUNK;
    if (i >= n) return;
    uint32_t lo = 0, hi = n_netile;                  // last non-empty tile whose region starts at or before i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)tile_off[netile[mid]] <= i) lo = mid; else hi = mid; }
    chunk[k] = (int32_t)lo - ((uint64_t)tile_off[netile[lo]] == i ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// k_resolve: one streaming pass over the tile index per count.  For every static entry: admission from the read's flag bits and MAPQ
// under THIS count's parameters, cell type from THIS barcode table; the admitted entries of a tile are written, order-preserving, as
// the walk's 8-byte records into the tile's region of cell type 0 or 1 (position = tile-relative exclusive count of that cell type: a
// segmented prefix sum over the static order), with the run flags the grouping kernels used to compute (first admitted entry of its
// barcode's run; run of exactly one entry).  The units' sizes fall out at every tile's last entry.  Workgroup = chunk of IX_CHUNK
// consecutive entries; a chunk that begins inside a tile takes the running counts from its predecessor's carry word (a chunk that
// contains a tile start publishes its own carry without waiting: only the chunks inside one deep tile form a chain).
struct RScan { uint32_t v, f; };          // v: c0 [0..11] | c1 [12..23] | admitted entry since the run start [24];  f: tile start seen [0] | run start seen [1] | tile starts [8..]
__device__ __forceinline__ RScan rs_combine(RScan a, RScan b) {      // a earlier, b later
    RScan r;
    const uint32_t cnt = (b.f & 1u) ? (b.v & 0xffffffu) : ((a.v + b.v) & 0xffffffu);
    const uint32_t run = (b.f & 2u) ? (b.v & (1u << 24)) : ((a.v | b.v) & (1u << 24));
    r.v = cnt | run;
    r.f = ((a.f | b.f) & 3u) | ((a.f & ~0xffu) + (b.f & ~0xffu));
    return r;
}
constexpr int RES_THREADS = 256, RES_PER = IX_CHUNK / RES_THREADS;
constexpr int IX_STAT_SLOTS = 256;

// the front half both passes share: the chunk's entries, their classes, the thread-sequential and workgroup-wide segmented scans
struct ResFront {
    uint32_t x0[RES_PER], x2[RES_PER + 1], cls[RES_PER];
    RScan incl[RES_PER];          // inclusive scan inside the thread
    RScan ex, all;                // everything of the chunk before this thread; the whole chunk
    unsigned long long ev, sg, ne;
};
__device__ __forceinline__ void resolve_front(const CountArgs& a, uint64_t k, int t, RScan* s_wave, ResFront& r) {
    const int lane = t & 63, wv = t >> 6;
    const uint64_t base = k * IX_CHUNK + (uint64_t)t * RES_PER, N = a.ix_n;
    {   // the index arrays are padded to whole chunks (k_ix_gather): 2 x 16 bytes per array and thread
        const uint4* p0 = reinterpret_cast<const uint4*>(a.ix0 + base);
        const uint4* p2 = reinterpret_cast<const uint4*>(a.ix2 + base);
        static_assert(RES_PER == 8, "two uint4 per thread");
        const uint4 a0 = p0[0], a1 = p0[1], c0 = p2[0], c1 = p2[1];
        r.x0[0] = a0.x; r.x0[1] = a0.y; r.x0[2] = a0.z; r.x0[3] = a0.w; r.x0[4] = a1.x; r.x0[5] = a1.y; r.x0[6] = a1.z; r.x0[7] = a1.w;
        r.x2[0] = c0.x; r.x2[1] = c0.y; r.x2[2] = c0.z; r.x2[3] = c0.w; r.x2[4] = c1.x; r.x2[5] = c1.y; r.x2[6] = c1.z; r.x2[7] = c1.w;
        r.x2[RES_PER] = a.ix2[base + RES_PER];
    }
    RScan run{0u, 0u};
    r.ev = 0; r.sg = 0; r.ne = 0;
#pragma unroll
    for (int q = 0; q < RES_PER; ++q) {
        const uint32_t flag = r.x2[q] & 0xfffu, mapq = (r.x2[q] >> 12) & 0xffu, cb = r.x0[q] & CB_MASK;
        bool ok = base + q < N && (flag & a.flag_exclude) == 0 && (int)mapq >= a.min_mq && cb < (uint32_t)a.n_cb;
        if (ok && a.ignore_orphans && (flag & 1u) && !(flag & 2u)) ok = false;
        uint32_t c = 2;          // cell type (0 / 1) or 2 = not counted
        if (ok) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct) c = ct; }
        r.cls[q] = c;
        if (c < 2) { r.ev += ((r.x0[q] >> 24) & 63u) + 1u; r.sg += (r.x2[q] & IX_SEGSTART) ? 1u : 0u; ++r.ne; }
        RScan e;
        e.v = (c == 0 ? 1u : 0u) | (c == 1 ? 1u << 12 : 0u) | (c < 2 ? 1u << 24 : 0u);
        e.f = ((r.x2[q] & IX_TILESTART) ? 0x101u : 0u) | ((r.x0[q] & IX_RUNSTART) ? 2u : 0u);
        run = q ? rs_combine(run, e) : e;
        r.incl[q] = run;
    }
    RScan sc = run;
    for (int o = 1; o < 64; o <<= 1) {
        RScan up; up.v = __shfl_up(sc.v, o); up.f = __shfl_up(sc.f, o);
        if (lane >= o) sc = rs_combine(up, sc);
    }
    if (lane == 63) s_wave[wv] = sc;
    RScan ex; ex.v = __shfl_up(sc.v, 1); ex.f = __shfl_up(sc.f, 1);
    if (lane == 0) { ex.v = 0; ex.f = 0; }
    __syncthreads();
    RScan wpre{0u, 0u};
    for (int w = 0; w < wv; ++w) wpre = w ? rs_combine(wpre, s_wave[w]) : s_wave[0];
    if (wv > 0) ex = lane == 0 ? wpre : rs_combine(wpre, ex);
    RScan all = s_wave[0];
    for (int w = 1; w < RES_THREADS / 64; ++w) all = rs_combine(all, s_wave[w]);
    r.ex = ex; r.all = all;
}
// aggregate word of a chunk: admitted entries of cell type 0 / 1 since the chunk's last tile start (or its beginning) [0..23] [24..47],
// an admitted entry since its last run start [48], a run start seen [49], a tile start seen [50]
constexpr unsigned long long IXA_RUN = 1ull << 48, IXA_RUNSTART = 1ull << 49, IXA_TILESTART = 1ull << 50;

// pass 1: every chunk's aggregate
__global__ __launch_bounds__(RES_THREADS) void k_resolve_agg(CountArgs a) {
    __shared__ RScan s_wave[RES_THREADS / 64];
    ResFront r;
    resolve_front(a, blockIdx.x, threadIdx.x, s_wave, r);
    if (threadIdx.x == 0)
        a.ix_carry[blockIdx.x] = (unsigned long long)(r.all.v & 0xfffu) | ((unsigned long long)((r.all.v >> 12) & 0xfffu) << 24) | ((r.all.v >> 24) & 1u ? IXA_RUN : 0ull) |
                                 ((r.all.f & 2u) ? IXA_RUNSTART : 0ull) | ((r.all.f & 1u) ? IXA_TILESTART : 0ull);
}

// pass 2
__global__ __launch_bounds__(RES_THREADS) void k_resolve(CountArgs a, unsigned long long* stat_slots) {
    __shared__ RScan s_wave[RES_THREADS / 64];
    __shared__ unsigned long long s_carry;
    __shared__ uint8_t s_first[RES_THREADS + 1];          // per thread: 0 no admitted entry, 1 its first admitted entry starts a run, 2 it does not
    __shared__ unsigned long long s_stat[3];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint64_t k = blockIdx.x, base = k * IX_CHUNK + (uint64_t)t * RES_PER;
    const uint64_t N = a.ix_n;
    ResFront r;
    resolve_front(a, k, t, s_wave, r);
    uint32_t (&x0)[RES_PER] = r.x0; uint32_t (&x2)[RES_PER + 1] = r.x2; uint32_t (&cls)[RES_PER] = r.cls; RScan (&incl)[RES_PER] = r.incl;
    const RScan ex = r.ex;
    unsigned long long ev = r.ev, sg = r.sg, ne = r.ne;
    // the running counts this chunk starts from: the aggregates of the chunks before it, back to the nearest one holding a tile start
    // (all final after pass 1: a look-back without waiting), 64 chunks per step
    if (wv == 0) {
        unsigned long long c0 = 0, c1 = 0, run = 0;
        bool run_closed = false;
        const bool need = k > 0 && !(x2[0] & IX_TILESTART);           // lane 0 of wave 0 holds the chunk's first entry
        bool go = __shfl((int)need, 0) != 0;
        int64_t j = (int64_t)k;
        while (go) {
            const int64_t idx = j - 1 - lane;
            const unsigned long long w = idx >= 0 ? a.ix_carry[idx] : IXA_TILESTART;
            const unsigned long long tmask = __ballot((w & IXA_TILESTART) != 0);
            const int nearest = tmask ? __ffsll((long long)tmask) - 1 : 63;
            const bool part = lane <= nearest;
            if (!run_closed) {
                const unsigned long long rmask = __ballot(part && (w & IXA_RUNSTART) != 0);
                const int rnear = rmask ? __ffsll((long long)rmask) - 1 : nearest;
                if (__ballot(part && lane <= rnear && (w & IXA_RUN) != 0)) run = 1;
                if (rmask) run_closed = true;
            }
            unsigned long long s0 = part ? (w & 0xffffffull) : 0ull, s1 = part ? ((w >> 24) & 0xffffffull) : 0ull;
            for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
            c0 += s0; c1 += s1;
            if (tmask) break;
            j -= 64;
        }
        if (lane == 0) { s_carry = c0 | (c1 << 24) | (run << 48); s_stat[0] = 0; s_stat[1] = 0; s_stat[2] = 0; }
    }
    __syncthreads();
    unsigned long long cin;
    cin = s_carry;
    const uint32_t cin0 = (uint32_t)(cin & 0xffffffull), cin1 = (uint32_t)((cin >> 24) & 0xffffffull), cinr = (uint32_t)((cin >> 48) & 1ull);
    // per entry: is it the first admitted entry of its run?
    bool newrun[RES_PER];
    uint8_t first_code = 0;
#pragma unroll
    for (int q = 0; q < RES_PER; ++q) {
        const RScan before = q ? rs_combine(ex, incl[q - 1]) : ex;                 // everything of the chunk before this entry
        uint32_t adm_before;
        if (x0[q] & IX_RUNSTART) adm_before = 0;
        else adm_before = (before.f & 2u) ? ((before.v >> 24) & 1u) : (((before.v >> 24) & 1u) | cinr);
        newrun[q] = cls[q] < 2 && !adm_before;
        if (cls[q] < 2 && !first_code) first_code = newrun[q] ? 1 : 2;
    }
    s_first[t] = first_code;
    if (t == 0) s_first[RES_THREADS] = 0;
    __syncthreads();
    const int32_t ctile = a.ix_chunk[k];
    const uint64_t ev_base = (uint64_t)(uintptr_t)a.events;
#pragma unroll
    for (int q = 0; q < RES_PER; ++q) {
        const uint64_t i = base + q;
        if (i >= N) break;
        const RScan upto = rs_combine(ex, incl[q]);                                  // the chunk up to and including this entry
        const uint32_t tile = a.ix_netile[ctile + (int32_t)(upto.f >> 8)];
        const bool in_region = tile >= a.tile_lo && tile < a.tile_hi;
        const uint32_t c0 = (upto.v & 0xfffu) + ((upto.f & 1u) ? 0u : cin0), c1 = ((upto.v >> 12) & 0xfffu) + ((upto.f & 1u) ? 0u : cin1);      // tile-relative, inclusive
        const uint32_t off = a.tile_off[tile], cap = a.tile_off[tile + 1] - off;
        const uint32_t b0 = a.n_ct == 1 ? off : 2u * off, b1 = 2u * off + cap;
        if (cls[q] < 2 && in_region) {
            // a run of exactly one entry: the next admitted entry of the chunk starts a run (conservative at the chunk's end)
            bool single = false;
            if (newrun[q]) {
                int nx = 0;
#pragma unroll
                for (int r = RES_PER - 1; r > q; --r) if (cls[r] < 2) nx = newrun[r] ? 1 : 2;
                if (!nx) { int tt = t + 1; while (tt < RES_THREADS && !s_first[tt]) ++tt; nx = s_first[tt]; }
                single = nx == 1;
            }
            const uint64_t addr = ev_base + ((uint64_t)a.ix1[i] << 7);
            const uint32_t meta = ((uint32_t)(addr >> 32) & 0x7fffu) | (x0[q] & 0x7f000000u) | (newrun[q] ? (single ? (META_NEWRUN | META_SINGLE) : META_NEWRUN) : 0u);
            const uint32_t pos = cls[q] == 0 ? b0 + c0 - 1u : b1 + c1 - 1u;
            a.rec[pos] = make_uint2((uint32_t)addr, meta);
        }
        if (x2[q + 1] & IX_TILESTART) {                                             // last static entry of its tile: the units' sizes and places
            if (in_region) {
                if (a.n_ct == 1) { a.unit_cnt[tile] = c0; a.unit_off[tile] = b0; }
                else { a.unit_cnt[2 * tile] = c0; a.unit_off[2 * tile] = b0; a.unit_cnt[2 * tile + 1] = c1; a.unit_off[2 * tile + 1] = b1; }
            }
        }
        if (cls[q] < 2 && !in_region) { ev -= ((x0[q] >> 24) & 63u) + 1u; sg -= (x2[q] & IX_SEGSTART) ? 1u : 0u; --ne; }
    }
    // statistics: admitted events and segments, spread over IX_STAT_SLOTS words (one word takes ~90 atomics per microsecond)
    for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
    if (lane == 0 && ne) { atomicAdd(&s_stat[0], ev); atomicAdd(&s_stat[1], sg); atomicAdd(&s_stat[2], ne); }
    __syncthreads();
    if (t == 0 && s_stat[2]) {
        unsigned long long* slot = stat_slots + (size_t)(blockIdx.x % IX_STAT_SLOTS) * 8;      // 64 bytes apart
        atomicAdd(&slot[0], s_stat[0]); atomicAdd(&slot[1], s_stat[1]); atomicAdd(&slot[2], s_stat[2]);
    }
}
__global__ void k_resolve_stats(CountArgs a, const unsigned long long* stat_slots) {
    unsigned long long ev = 0, sg = 0, ne = 0;
    for (int i = threadIdx.x; i < IX_STAT_SLOTS; i += blockDim.x) { ev += stat_slots[(size_t)i * 8]; sg += stat_slots[(size_t)i * 8 + 1]; ne += stat_slots[(size_t)i * 8 + 2]; }
    for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
    __shared__ unsigned long long s[3];
    if (threadIdx.x == 0) { s[0] = 0; s[1] = 0; s[2] = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s[0], ev); atomicAdd(&s[1], sg); atomicAdd(&s[2], ne); }
    __syncthreads();
    if (threadIdx.x == 0) { a.scalars[SC_EVENTS] = s[0]; a.scalars[SC_SEGS] = s[1]; a.scalars[SC_NENT] = s[2]; }
}

// Work lists of the deep and the mid units when their records arrive grouped (tile index): a big slot j of a unit of n records and
// nsub slots is [cut(n j / nsub), cut(n (j + 1) / nsub)) with cut(x) = the first record at or after x that starts a barcode run (or n),
// its NSLICE run-aligned slices likewise.  One thread per (big slot, cut); every cut is found on its own by walking forward over at
// most one run.
__global__ void k_cut(CountArgs a) {
    const uint32_t n_big = a.n_slots - (uint32_t)a.scalars[SC_NSMALL];
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t qi = id / (NSLICE + 1), q = id % (NSLICE + 1);
    if (qi >= n_big) return;
    const uint32_t s = a.slot_list[a.n_slots - 1 - qi];
    const uint32_t w = a.slot_w[s], u = a.ne_units[w];
    const uint32_t n = a.unit_cnt[u], uoff = a.unit_off[u], nsub = a.ne_nslot[w], j = s - a.ne_slot_base[w];
    auto cut = [&](uint32_t x) -> uint32_t { while (x < n && !(a.rec[uoff + x].y & META_NEWRUN)) ++x; return x < n ? x : n; };
    const uint32_t lo = j == 0 ? 0u : cut((uint32_t)(((uint64_t)n * j) / nsub));
    const uint32_t hi = j + 1 == nsub ? n : cut((uint32_t)(((uint64_t)n * (j + 1)) / nsub));
    uint32_t b;
    if (q == 0) b = lo; else if (q == NSLICE) b = hi;
    else { b = cut(lo + (uint32_t)(((uint64_t)(hi - lo) * q) / NSLICE)); if (b > hi) b = hi; if (b < lo) b = lo; }
    a.slices[(uint64_t)s * (NSLICE + 1) + q] = b - lo;
    if (q == 0) { a.slot_cnt[s] = hi - lo; a.slot_off[s] = uoff + lo; }
}

static int build_index(lsg_ctx* c) {
    if (c->index_valid) return 0;
    if (tile_capacities(c)) return -1;
    hipStream_t st = c->stream;
    const int64_t S = c->rd.n_segs;
    // the build's temporaries live in the context (grow-only, like every other workspace): allocating and freeing ~9 GB per build costs
    // more wall time (0.9 s) than the build's kernels (60 ms)
    DevBuf &key_a = c->bt[0], &key_b = c->bt[1], &val_a = c->bt[2], &val_b = c->bt[3], &eread = c->bt[4], &tmp = c->bt[5];
    auto done = [&](int rc) { return rc; };
    uint32_t total = 0, max_cap = 0;
    {   // k_resolve carries a tile's running counts in 24-bit fields: a tile of 2^24 entries or more leaves the counts to the scatter path
        uint32_t* d_max = reinterpret_cast<uint32_t*>(c->d_scalars.as<unsigned long long>() + SC_NNE);
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceReduce::Max(nullptr, tb, c->d_tile_cap.as<uint32_t>(), d_max, (int)c->n_tiles, st));
        if (tmp.reserve(tb + 256)) return done(-1);
        tb = tmp.cap;
        LSG_HIP(hipcub::DeviceReduce::Max(tmp.p, tb, c->d_tile_cap.as<uint32_t>(), d_max, (int)c->n_tiles, st));
        LSG_HIP(hipMemcpyAsync(&max_cap, d_max, 4, hipMemcpyDeviceToHost, st));
    }
    LSG_HIP(hipMemcpyAsync(&total, c->d_tile_off.as<uint32_t>() + c->n_tiles, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    const uint64_t N = max_cap < (1u << 24) ? total : 0;
    c->ix_n = N; c->ix_n_netile = 0;
    if (N == 0 || S == 0) { c->index_valid = true; return done(0); }
    if (key_a.reserve(N * 8) || key_b.reserve(N * 8) || val_a.reserve(N * 4) || val_b.reserve(N * 4) || eread.reserve(N * 4) ||
        c->ws[WS_ENT].reserve((c->entries_upper + 1) * 16 + 64) || c->ws[WS_SEG_INFO].reserve(((size_t)S + 1) * 8) ||
        c->d_ix0.reserve((N + IX_CHUNK + 16) * 4) || c->d_ix1.reserve((N + IX_CHUNK + 16) * 4) || c->d_ix2.reserve((N + IX_CHUNK + 16) * 4) ||
        c->d_ix_netile.reserve(((size_t)c->n_tiles + 2) * 4) || c->d_ix_chunk.reserve((N / IX_CHUNK + 2) * 4) || c->d_ix_carry.reserve((N / IX_CHUNK + 2) * 8))
        return done(-1);
    {   // the scatter, as ONE cell type over every read with a barcode
        LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, st));
        CountArgs a{};
        a.n_reads = c->rd.n_reads; a.n_segs = S;
        a.read_tid = c->rd.read_tid; a.read_cb = c->rd.read_cb; a.read_flag = c->rd.read_flag;
        a.seg_read = c->rd.seg_read; a.seg_start = c->rd.seg_start; a.seg_len = c->rd.seg_len; a.seg_ev_off = c->rd.seg_ev_off;
        a.tile_base = c->d_tile_base.as<uint32_t>(); a.contig_len = c->d_contig_len.as<int64_t>(); a.n_contigs = c->n_contigs;
        a.n_ct = 1; a.tile_lo = 0; a.tile_hi = c->n_tiles; a.two_ended = 1;
        a.seg_info = c->ws[WS_SEG_INFO].as<uint2>(); a.ent = c->ws[WS_ENT].as<uint2>();
        a.tile_off = c->d_tile_off.as<uint32_t>(); a.cur_lo = c->d_cur_lo.as<uint32_t>(); a.cur_hi = c->d_cur_hi.as<uint32_t>();
        a.scalars = c->d_scalars.as<unsigned long long>();
        a.ixb_key = key_a.as<uint64_t>(); a.ixb_read = eread.as<uint32_t>();
        LSG_HIP(hipMemcpyAsync(a.cur_lo, a.tile_off, (size_t)c->n_tiles * 4, hipMemcpyDeviceToDevice, st));
        unsigned g = (unsigned)((S + 255) / 256); if (g > (unsigned)(c->n_cus * 16)) g = (unsigned)(c->n_cus * 16);
        hipLaunchKernelGGL(k_seg_info_static, dim3(g), dim3(256), 0, st, a);
        unsigned seg_grid = (unsigned)((S + 256 * BIN_SUPER - 1) / (256 * BIN_SUPER));
        if (seg_grid > (unsigned)(c->n_cus * 8)) seg_grid = (unsigned)(c->n_cus * 8);
        hipLaunchKernelGGL(k_bin_segments<2>, dim3(seg_grid), dim3(256), 0, st, a);
    }
    {   // permutation that orders the entries by (tile, barcode)
        hipcub::CountingInputIterator<uint32_t> iota(0);
        int tile_bits = 1; while ((1ull << tile_bits) < (uint64_t)c->n_tiles + 1) ++tile_bits;
        LSG_HIP(hipMemsetAsync(val_a.p, 0, 4, st));
        {   // val_a = 0, 1, 2, ...
            size_t tb = 0;
            LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, hipcub::ConstantInputIterator<uint32_t>(1u), val_a.as<uint32_t>(), (int)N, st));
            if (tmp.reserve(tb + 256)) return done(-1);
            tb = tmp.cap;
            LSG_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, hipcub::ConstantInputIterator<uint32_t>(1u), val_a.as<uint32_t>(), (int)N, st));
        }
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key_a.as<uint64_t>(), key_b.as<uint64_t>(), val_a.as<uint32_t>(), val_b.as<uint32_t>(), (int)N, 0, 24 + tile_bits, st));
        if (tmp.reserve(tb + 256)) return done(-1);
        tb = tmp.cap;
        LSG_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, key_a.as<uint64_t>(), key_b.as<uint64_t>(), val_a.as<uint32_t>(), val_b.as<uint32_t>(), (int)N, 0, 24 + tile_bits, st));
    }
    hipLaunchKernelGGL(k_ix_gather, dim3((unsigned)(c->n_cus * 16)), dim3(256), 0, st, key_b.as<uint64_t>(), val_b.as<uint32_t>(), c->ws[WS_ENT].as<uint2>(), eread.as<uint32_t>(),
                       c->rd.read_flag, c->rd.read_mapq, N, c->d_ix0.as<uint32_t>(), c->d_ix1.as<uint32_t>(), c->d_ix2.as<uint32_t>());
    {   // the tiles that hold entries, in order; every chunk's first tile
        hipcub::CountingInputIterator<uint32_t> tile_it(0);
        CapNonZero pred{c->d_tile_cap.as<uint32_t>()};
        uint32_t* d_n = reinterpret_cast<uint32_t*>(c->d_scalars.as<unsigned long long>() + SC_NNE);
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceSelect::If(nullptr, tb, tile_it, c->d_ix_netile.as<uint32_t>(), d_n, (int)c->n_tiles, pred, st));
        if (tmp.reserve(tb + 256)) return done(-1);
        tb = tmp.cap;
        LSG_HIP(hipcub::DeviceSelect::If(tmp.p, tb, tile_it, c->d_ix_netile.as<uint32_t>(), d_n, (int)c->n_tiles, pred, st));
        LSG_HIP(hipMemcpyAsync(&c->ix_n_netile, d_n, 4, hipMemcpyDeviceToHost, st));
        LSG_HIP(hipStreamSynchronize(st));
        const uint64_t n_chunks = (N + IX_CHUNK - 1) / IX_CHUNK;
        hipLaunchKernelGGL(k_ix_chunks, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, c->d_ix_netile.as<uint32_t>(), c->ix_n_netile,
                           c->d_tile_off.as<uint32_t>(), N, c->d_ix_chunk.as<int32_t>());
    }
    LSG_HIP(hipGetLastError());
    LSG_HIP(hipStreamSynchronize(st));
    c->index_valid = true;
    return done(0);
}

// ================================================================================================
// Tile-major store.  Built once per (load, read filters, number of cell types) on top of the tile index: the admitted entries' events
// are copied out of the read-major lines into the index order (tile, barcode), eight entries to a 1 KB block held TRANSPOSED
// ([position 0..63][entry 0..7], 16 bytes per position), every tile padded to whole blocks.  A count then streams each tile's blocks
// front to back: one 16-byte load per lane brings the lane's position of eight entries, a kilobyte per wave instruction instead of the
// 128-byte gathers of the read-major layout (which stop at ~4 TB/s whatever their shape; contiguous kilobytes reach ~6).  What a count
// still decides per entry is the cell type of its barcode (k_tm_resolve: one byte per entry); both cell types of a tile are counted in
// the same pass (a barcode's run belongs to one cell type), into two pairs of LDS planes per wave.
//   s0[p]   cb [0..23] | forward << 30 | first entry of its barcode's run in the tile << 31          (pad entries: cb = CB_MASK, run start)
//   b[p]    events - 1 [0..5] | first line of its segment << 6 | run of exactly one entry << 7
//   meta[p] (per count, 32 bits laid out so that the walk uses them as operands): cell type << 4 and << 12 | forward << 20 |
//           not counted or not there << 29 | run of one entry << 30 | run start << 31
// Tiles of more than TM_JOB_TGT entries are cut at run starts into jobs of about TM_JOB_TGT entries (one wave each, partial sums to
// slabs that k_finalize_multi adds up: 8 KB per job and cell type, so jobs are as long as the planes' fields allow); everything about jobs, units and slabs is static too, so a count has no planning step and
// one host synchronisation (its final read of the counters).
#ifndef LSG_TM_ASM
#define LSG_TM_ASM true
#endif
constexpr int TM_JOB_TGT = 3072, TM_JOB_LIMIT = 4095;      // LIMIT: what the planes' 12-bit forward field holds; a cut moves forward to the next run start
constexpr uint32_t TM_PAD_S0 = CB_MASK | IX_RUNSTART;
enum { TM_STORE = 0, TM_S0, TM_B, TM_LINE, TM_META, TM_BLK_TILE, TM_JOBS, TM_NE_UNITS, TM_NE_GEOM, TM_NE_NSLOT, TM_NE_ACC, TM_MULTI, TM_CHUNKS, TM_EXT, TM_NBUF };
constexpr uint32_t TM_CHUNK_WORK = 4096, TM_JOB_W0 = 32;      // a workgroup dequeues at most this much work (entries + a constant per job) at a time
struct TmJob { uint32_t e0, e1, w0, slab, nj, cnt, tile, emid; };     // padded-entry range; unit of (tile, cell type 0); slab of (job, cell type 0) or ~0; jobs and entries of the tile; where the job's second wave starts (a run start, or e1)
constexpr uint32_t TMM_CT4 = 1u << 4, TMM_CT12 = 1u << 12, TMM_FWD = 1u << 20, TMM_SKIP = 1u << 29, TMM_SINGLE = 1u << 30, TMM_RS = 1u << 31;
struct TmArgs {
    const uint4* store; const uint32_t* s0; const uint8_t* b; uint32_t* meta; const uint32_t* blk_tile; const TmJob* jobs;
    const uint32_t* chunk_start;          // static: first job of every chunk of about TM_CHUNK_WORK work
    const uint16_t* ext;                  // static, per block: first position any of its entries has an event at | one past the last << 8
    uint64_t np; uint32_t nblk, njobs, nchunks;
};

struct TmAdm {      // an index entry passes the key's read filters (ix2: flag12 | mapq << 12 | ...)
    const uint32_t* ix2; uint32_t flag_exclude; int32_t min_mq, ignore_orphans;
    __host__ __device__ uint32_t operator()(const uint32_t& i) const {
        const uint32_t x = ix2[i], flag = x & 0xfffu;
        bool ok = (flag & flag_exclude) == 0 && (int)((x >> 12) & 0xffu) >= min_mq;
        if (ok && ignore_orphans && (flag & 1u) && !(flag & 2u)) ok = false;
        return ok ? 1u : 0u;
    }
};
// per tile: admitted entries, blocks, non-empty, jobs, slabs, multi-job (inputs of five exclusive scans)
__global__ void k_tm_tiles(const uint32_t* tile_off, const uint32_t* S, uint32_t n_tiles, int n_ct, uint32_t job_tgt, uint32_t* cnt, uint32_t* blk, uint32_t* ne,
                           uint32_t* nj, uint32_t* slabs, uint32_t* multi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint32_t c = 0;
    if (t < n_tiles) c = S[tile_off[t + 1]] - S[tile_off[t]];
    const uint32_t j = c == 0 ? 0u : (c <= job_tgt ? 1u : (c + job_tgt - 1) / job_tgt);
    cnt[t] = c; blk[t] = (c + 7) / 8; ne[t] = c ? 1u : 0u; nj[t] = j; slabs[t] = j > 1 ? j * (uint32_t)n_ct : 0u; multi[t] = j > 1 ? 1u : 0u;
}
// largest t in [0, n) with off[t] <= x (off non-decreasing, off[0] <= x): the tile whose region holds x, skipping empty ones
__device__ __forceinline__ uint32_t tm_owner(const uint32_t* off, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
__global__ void k_tm_blk_tile(const uint32_t* blk_off, uint32_t n_tiles, uint32_t nblk, uint32_t* blk_tile) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) blk_tile[b] = tm_owner(blk_off, n_tiles, b);
}
__global__ void k_tm_fill(TmAdm adm, const uint32_t* ix0, const uint32_t* ix1, const uint32_t* ix2, uint64_t n, const uint32_t* tile_off, uint32_t n_tiles,
                          const uint32_t* S, const uint32_t* blk_off, uint32_t* s0, uint32_t* line, uint8_t* b) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!adm((uint32_t)i)) continue;
        const uint32_t t = tm_owner(tile_off, n_tiles, (uint32_t)i);
        const uint64_t p = (uint64_t)blk_off[t] * 8 + (S[i] - S[tile_off[t]]);
        const uint32_t x = ix0[i];
        s0[p] = x & (CB_MASK | META_FWD);
        line[p] = ix1[i];
        b[p] = (uint8_t)(((x >> 24) & 63u) | ((ix2[i] & IX_SEGSTART) ? 64u : 0u));
    }
}
__device__ __forceinline__ bool tm_first_of_tile(uint64_t p, const uint32_t* blk_off, const uint32_t* blk_tile) {
    return (p & 7) == 0 && (uint64_t)blk_off[blk_tile[p >> 3]] * 8 == p;
}
// run flags over the admitted entries: first entry of its barcode in the tile; run of exactly one
__global__ void k_tm_runs(uint32_t* s0, uint8_t* b, uint64_t np, const uint32_t* blk_off, const uint32_t* blk_tile) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = s0[p], cb = x & CB_MASK;
        if (cb == CB_MASK) continue;                                   // pad: a run start already
        const bool rs = tm_first_of_tile(p, blk_off, blk_tile) || (s0[p - 1] & CB_MASK) != cb;
        const bool next_rs = p + 1 >= np || (s0[p + 1] & CB_MASK) != cb || tm_first_of_tile(p + 1, blk_off, blk_tile);
        if (rs) s0[p] = x | IX_RUNSTART;                               // (neighbours read bits 0..23 only)
        if (rs && next_rs) b[p] |= 128u;
    }
}
// one wave per block: lane = position; eight 128-byte lines in, one transposed kilobyte out
constexpr int TMG_BLOCKS = 4;          // blocks per wave: 32 line loads in flight
__global__ void k_tm_gather(const uint16_t* events, const uint32_t* s0, const uint32_t* line, uint32_t nblk, uint4* store, uint16_t* ext) {
    const int lane = threadIdx.x & 63;
    const uint32_t blk0 = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * TMG_BLOCKS;
    if (blk0 >= nblk) return;
    uint32_t e[TMG_BLOCKS][8];
#pragma unroll
    for (int q = 0; q < TMG_BLOCKS; ++q) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t p = (uint64_t)(blk0 + q) * 8 + u;                    // (the entry arrays are padded past the last block)
            e[q][u] = blk0 + q < nblk && (s0[p] & CB_MASK) != CB_MASK ? (uint32_t)events[(uint64_t)line[p] * 64 + lane] : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < TMG_BLOCKS; ++q) {
        const uint32_t blk = blk0 + q;
        if (blk >= nblk) break;
        uint32_t any = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) any |= e[q][u];
        store[(uint64_t)blk * 64 + lane] = make_uint4(e[q][0] | (e[q][1] << 16), e[q][2] | (e[q][3] << 16), e[q][4] | (e[q][5] << 16), e[q][6] | (e[q][7] << 16));
        // the positions outside [first, last] of the block's events are zeros in all eight entries (exon and read ends shared by the
        // tile's reads): the walk does not fetch them (its buffer descriptor ends there, lanes outside read zeros)
        const unsigned long long m = __ballot(any != 0u);
        if (lane == 0) ext[blk] = m ? (uint16_t)(__ffsll((long long)m) - 1) | (uint16_t)((64 - __clzll((long long)m)) << 8) : (uint16_t)0;
    }
}
// per non-empty tile: its units (one per cell type), its jobs cut at run starts
__global__ void k_tm_jobs(CountArgs a, const uint32_t* s0, const uint32_t* cnt, const uint32_t* blk_off, const uint32_t* ne_off, const uint32_t* nj,
                          const uint32_t* job_off, const uint32_t* slab_off, const uint32_t* multi_off, uint32_t n_tiles, TmJob* jobs,
                          uint32_t* ne_units, int2* ne_geom, uint32_t* ne_nslot, uint32_t* ne_acc, uint32_t* multi, uint32_t* max_job) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t n = cnt[t];
    if (!n) return;
    const uint32_t J = nj[t], ord = ne_off[t];
    const uint64_t base = (uint64_t)blk_off[t] * 8;
    for (int ct = 0; ct < a.n_ct; ++ct) {
        const uint32_t w = ord * (uint32_t)a.n_ct + ct, u = t * (uint32_t)a.n_ct + ct;
        ne_units[w] = u;
        int c2, tid; int32_t tstart;
        unit_geometry(a, u, c2, tid, tstart);
        ne_geom[w] = make_int2(tstart, tid | (ct << 24));
        ne_nslot[w] = J;
        ne_acc[w] = J > 1 ? slab_off[t] + (uint32_t)ct * J : 0u;
        if (J > 1) multi[multi_off[t] * (uint32_t)a.n_ct + ct] = w;
    }
    auto cut = [&](uint32_t x) -> uint32_t { while (x < n && !(s0[base + x] & IX_RUNSTART)) ++x; return x < n ? x : n; };
    uint32_t e0 = 0;
    for (uint32_t j = 0; j < J; ++j) {
        const uint32_t e1 = j + 1 == J ? n : cut((uint32_t)(((uint64_t)n * (j + 1)) / J));
        TmJob jb;
        jb.e0 = (uint32_t)(base + e0); jb.e1 = (uint32_t)(base + (e1 < e0 ? e0 : e1)); jb.w0 = ord * (uint32_t)a.n_ct;
        jb.slab = J > 1 ? slab_off[t] + j : 0xFFFFFFFFu; jb.nj = J; jb.cnt = n; jb.tile = t;
        {   // two waves share the job: the second starts at the run start at or after its middle (short jobs: one wave)
            const uint32_t a0 = e0, a1 = e1 < e0 ? e0 : e1;
            uint32_t mid = a1;
            if (a1 - a0 >= 64u) { mid = cut(a0 + (a1 - a0) / 2u); if (mid > a1) mid = a1; }
            jb.emid = (uint32_t)(base + mid);
        }
        jobs[job_off[t] + j] = jb;
        if (jb.e1 - jb.e0 > (uint32_t)TM_JOB_LIMIT) atomicMax(max_job, jb.e1 - jb.e0);
        e0 = e1 < e0 ? e0 : e1;
    }
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

// per count: the byte the walk reads per entry
__global__ __launch_bounds__(256) void k_tm_resolve(CountArgs a, TmArgs tm, unsigned long long* stat_slots) {
    __shared__ unsigned long long s_stat[3];
    if (threadIdx.x == 0) { s_stat[0] = 0; s_stat[1] = 0; s_stat[2] = 0; }
    __syncthreads();
    unsigned long long ev = 0, sg = 0, ne = 0;
    const uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk < tm.nblk) {
        const uint32_t tile = tm.blk_tile[blk];
        const bool in_region = tile >= a.tile_lo && tile < a.tile_hi;
        const uint4* sp = reinterpret_cast<const uint4*>(tm.s0 + (uint64_t)blk * 8);
        const uint4 s_lo = sp[0], s_hi = sp[1];
        const uint2 bb = *reinterpret_cast<const uint2*>(tm.b + (uint64_t)blk * 8);
        const uint32_t sv[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
        uint32_t mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t s = sv[u], b8 = ((u < 4 ? bb.x : bb.y) >> (8 * (u & 3))) & 0xffu, cb = s & CB_MASK;
            uint32_t cls = 3;
            if (in_region) {
                cls = 2;
                if (cb < (uint32_t)a.n_cb) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct) cls = ct; }
            }
            if (cls < 2) { ev += (b8 & 63u) + 1u; sg += (b8 >> 6) & 1u; ++ne; }
            uint32_t m = cls < 2 ? (cls ? (TMM_CT4 | TMM_CT12) : 0u) | ((s & META_FWD) ? TMM_FWD : 0u) | ((b8 & 128u) ? TMM_SINGLE : 0u) : TMM_SKIP;
            if (cls != 3 && (s & IX_RUNSTART)) m |= TMM_RS;              // (an entry that is not there starts nothing)
            mv[u] = m;
        }
        uint4* mp = reinterpret_cast<uint4*>(tm.meta + (uint64_t)blk * 8);
        mp[0] = make_uint4(mv[0], mv[1], mv[2], mv[3]); mp[1] = make_uint4(mv[4], mv[5], mv[6], mv[7]);
    }
    for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
    if ((threadIdx.x & 63) == 0 && ne) { atomicAdd(&s_stat[0], ev); atomicAdd(&s_stat[1], sg); atomicAdd(&s_stat[2], ne); }
    __syncthreads();
    if (threadIdx.x == 0 && s_stat[2]) {
        unsigned long long* slot = stat_slots + (size_t)(blockIdx.x % IX_STAT_SLOTS) * 8;
        atomicAdd(&slot[0], s_stat[0]); atomicAdd(&slot[1], s_stat[1]); atomicAdd(&slot[2], s_stat[2]);
    }
}

// run state of a wave over a tile's entries, both cell types.  nc packs the runs that counted an event per cell type (16 bits each);
// mask: bits 8..15 = symbols seen in the open run at this lane, bit 0 / bit 16 = the run (of cell type 0 / 1) counted an event here
struct TmState { uint32_t nc, mask; };
__device__ __forceinline__ void tm_lds_add(uint32_t addr, uint32_t v) {
    __hip_atomic_fetch_add((LSG_AS3 uint32_t*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// What happens to the run state at an entry depends on the meta words before it only, so a wave works it out for a whole group of
// entries at once (one lane per entry, tm_walk_range) and hands every entry its verdict in two more bits of its meta word:
constexpr uint32_t TMM_CLOSE = 1u << 27;      // a run of several entries is open when this run start arrives: close it first
constexpr uint32_t TMM_FIRST = 1u << 28;      // first entry that is there of a run of several (else: compare with the run's mask)
// One entry at the lane's position.  m: the entry's meta word in an SGPR (TMM_*); evw: the register holding the lane's events of an
// entry pair, HI picks the half.  Planes of cell type c at pkl0 + 4096 c: plane 0 quality sum [0..19] | forward count [20..31], plane 1
// (+2048) count [0..15] | duplicates [16..31].  Every decision is a scalar branch on a bit of m; the lanes that count the event are
// selected with EXEC (all lanes are active around the block).  A run of one entry: 7 vector operations; first entry of a longer
// run 8 (+3 when it closes the run before it); others 10.
constexpr bool TM_ASM = LSG_TM_ASM;
template <bool HI>
__device__ __forceinline__ void tm_add(TmState& s, uint32_t m, uint32_t evw, uint32_t thr, uint32_t pkl0, uint32_t one) {
    if (!TM_ASM) {                                   // the same in plain C++ (what the asm block is checked against when it is touched)
        const uint32_t ev = HI ? evw >> 16 : evw & 0xffffu;
        if (m & TMM_CLOSE) { s.nc += s.mask & 0x10001u; s.mask = 0; }
        if (m & TMM_SKIP) return;
        const bool counted = (ev & 0x8ffu) >= thr;
        const uint32_t addr = (pkl0 | (ev & 0x700u)) + (m & TMM_CT12);
        const uint32_t lo = (ev & 0xffu) | (m & TMM_FWD);
        const uint32_t sym8 = (ev >> 8) & 15u, ctone = 1u << (m & TMM_CT4);
        if (m & TMM_SINGLE) {
            if (counted) { tm_lds_add(addr, lo); tm_lds_add(addr + 2048u, 1u); s.nc += ctone; }
        } else if (m & TMM_FIRST) {
            if (counted) { tm_lds_add(addr, lo); tm_lds_add(addr + 2048u, 1u); s.mask = (1u << sym8) | ctone; }
        } else if (counted) {
            const uint32_t seen = (s.mask >> sym8) & 1u;
            tm_lds_add(addr, lo); tm_lds_add(addr + 2048u, 1u | (seen << 16));
            s.mask |= (1u << sym8) | ctone;
        }
        return;
    }
    uint32_t t0, t1, addr, lo, sa, sb;
#define LSG_TM_ADD(WSEL, BSEL, SYMPOS)                                                                                               \
    asm volatile(                                                                                                                    \
        "s_bitcmp1_b32 %[m], 27\n\t"                                                                                                  \
        "s_cbranch_scc0 1f\n\t"                                                                                                       \
        "v_and_b32 %[t1], 0x10001, %[mask]\n\t"                     /* close the run of several entries before this one */          \
        "v_add_u32 %[nc], %[nc], %[t1]\n\t"                                                                                           \
        "v_mov_b32 %[mask], 0\n"                                                                                                      \
        "1:\n\t"                                                                                                                      \
        "s_bitcmp1_b32 %[m], 29\n\t"                                                                                                  \
        "s_cbranch_scc1 5f\n\t"                                   /* not counted / not there */                                     \
        "v_and_b32_sdwa %[t0], %[k8ff], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" WSEL "\n\t"               \
        "s_and_b32 %[sa], %[m], 0x1000\n\t"                                                                                           \
        "v_and_b32_sdwa %[addr], %[c700], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" WSEL "\n\t"             \
        "s_and_b32 %[sb], %[m], 0x100000\n\t"                                                                                         \
        "v_or3_b32 %[addr], %[addr], %[pkl], %[sa]\n\t"             /* symbol row | lane word | the cell type's planes */           \
        "v_or_b32_sdwa %[lo], %[sb], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BSEL "\n\t"                  \
        "s_and_b32 %[sa], %[m], 16\n\t"                                                                                               \
        "v_cmpx_le_u32 vcc, %[thr], %[t0]\n\t"                      /* EXEC = the lanes that count this event */                    \
        "s_lshl_b32 %[sa], 1, %[sa]\n\t"                            /* bit 0 or bit 16: the cell type's run counter */              \
        "s_bitcmp1_b32 %[m], 30\n\t"                                                                                                  \
        "s_cbranch_scc0 2f\n\t"                                                                                                       \
        "ds_add_u32 %[addr], %[lo]\n\t"                             /* a run of one entry */                                        \
        "ds_add_u32 %[addr], %[one] offset:2048\n\t"                                                                                  \
        "v_add_u32 %[nc], %[sa], %[nc]\n\t"                                                                                           \
        "s_branch 4f\n"                                                                                                               \
        "2:\n\t"                                                                                                                      \
        "v_bfe_u32 %[t1], %[ev], " SYMPOS ", 4\n\t"                 /* 8 + class */                                                 \
        "s_bitcmp1_b32 %[m], 28\n\t"                                                                                                  \
        "s_cbranch_scc0 3f\n\t"                                                                                                       \
        "ds_add_u32 %[addr], %[lo]\n\t"                             /* first entry of a longer run that is there */                 \
        "ds_add_u32 %[addr], %[one] offset:2048\n\t"                                                                                  \
        "v_lshl_or_b32 %[mask], %[one], %[t1], %[sa]\n\t"                                                                             \
        "s_branch 4f\n"                                                                                                               \
        "3:\n\t"                                                                                                                      \
        "v_bfe_u32 %[t0], %[mask], %[t1], 1\n\t"                    /* symbol already seen in this run: duplicate */                \
        "v_lshl_or_b32 %[t0], %[t0], 16, %[one]\n\t"                                                                                  \
        "ds_add_u32 %[addr], %[lo]\n\t"                                                                                               \
        "ds_add_u32 %[addr], %[t0] offset:2048\n\t"                                                                                   \
        "v_lshl_or_b32 %[t1], %[one], %[t1], %[sa]\n\t"                                                                               \
        "v_or_b32 %[mask], %[mask], %[t1]\n"                                                                                          \
        "4:\n\t"                                                                                                                      \
        "s_mov_b64 exec, -1\n"                                                                                                        \
        "5:"                                                                                                                          \
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [addr] "=&v"(addr), [lo] "=&v"(lo), [sa] "=&s"(sa), [sb] "=&s"(sb),                        \
          [mask] "+v"(s.mask), [nc] "+v"(s.nc)                                                                                       \
        : [ev] "v"(evw), [m] "s"(m), [thr] "s"(thr), [c700] "s"(0x700u), [k8ff] "s"(0x8ffu), [one] "v"(one), [pkl] "v"(pkl0)           \
        : "scc", "vcc", "memory")
    if (HI) LSG_TM_ADD("WORD_1", "BYTE_2", "24"); else LSG_TM_ADD("WORD_0", "BYTE_0", "8");
#undef LSG_TM_ADD
}
struct TmCounters {
    const uint32_t* pl; int lane; uint32_t ncdup;
    __device__ __forceinline__ uint32_t BC(int k) const { return pl[512 + k * 64 + lane] & 0xffffu; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return pl[512 + k * 64 + lane] >> 16; }
    __device__ __forceinline__ uint32_t BQ(int k) const { return pl[k * 64 + lane] & 0xfffffu; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return pl[k * 64 + lane] >> 20; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};

constexpr int TMW_WAVES = 2, TM_GROUP = 4;       // two waves share a job (and its planes); blocks per load group: 4 KB in flight per wave and group
typedef uint32_t tm_u32x4 __attribute__((ext_vector_type(4)));
// entries [s0, s1) of the store walked by one wave into the planes at pkl0
__device__ __forceinline__ void tm_walk_range(const TmArgs& tm, TmState& st, uint32_t s0, uint32_t s1, uint32_t thr, uint32_t pkl0, uint32_t one, int lane) {
    const uint32_t b0 = s0 >> 3, nblk = ((s1 + 7) >> 3) - b0;
    const uint32_t* mp = tm.meta + (uint64_t)b0 * 8;
    const uint16_t* xp = tm.ext + b0;
    const uint64_t sbase = (uint64_t)(uintptr_t)(tm.store + (uint64_t)b0 * 64);
    const uint64_t sb = ((uint64_t)rl((uint32_t)(sbase >> 32), 0) << 32) | rl((uint32_t)sbase, 0);
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const int ng = (int)((nblk + TM_GROUP - 1) / TM_GROUP);
    const uint32_t pe0 = s0 - b0 * 8u, pe1 = s1 - b0 * 8u;                // the range, relative to its first block
    // a group = TM_GROUP blocks: 16 bytes per lane and block of events, and the meta words of its 8 TM_GROUP entries one per lane
    // (read back lane by lane into an SGPR when the entry's turn comes); the entries of the neighbouring ranges are not there.
    // Every block is fetched through a descriptor of its own that covers the positions its entries have events at (ext): the lanes
    // outside it read zeros without a memory request.  The extents of the NEXT group's blocks travel one group ahead, in lanes 0..3.
    tm_u32x4 EA[TM_GROUP], EB[TM_GROUP];
    uint32_t MA, MB;
    auto extents = [&](int g) -> uint32_t {
        const uint32_t blk = (uint32_t)g * TM_GROUP + (uint32_t)lane;
        return lane < TM_GROUP && blk < nblk ? (uint32_t)xp[blk] : 0u;
    };
    uint32_t X = extents(0);
    auto issue = [&](int g, tm_u32x4 (&E)[TM_GROUP], uint32_t& M) {
        const uint32_t Xg = X;
        X = extents(g + 1);
#pragma unroll
        for (int k = 0; k < TM_GROUP; ++k) {
            const uint32_t x = rl(Xg, k), plo = x & 0xffu, phi = x >> 8;
            const uint64_t base = sb + (uint64_t)((uint32_t)g * TM_GROUP + k) * 1024u + plo * 16u;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)base), 0, (int)((phi - plo) * 16u), 0x00020000);
            E[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane16 - plo * 16u), 0, 0);       // before plo: the offset wraps, out of range = zeros
        }
        const uint32_t pr = (uint32_t)g * (8u * TM_GROUP) + (uint32_t)lane;
        M = TMM_SKIP;
        if (lane < 8 * TM_GROUP && pr >= pe0 && pr < pe1) M = mp[pr];
    };
    static_assert(8 * TM_GROUP == 32, "one 32-bit mask per group");
    uint32_t open_in = 0;                            // wave-uniform: a run of several entries is open when the group begins
    // the run state before every entry of the group, from the meta words alone: entry i sees an open run iff an entry that is there
    // and not a run of one lies between the last run start before i (inclusive) and i - or none starts in the group and one was open
    auto verdicts = [&](uint32_t M) -> uint32_t {
        const bool there = !(M & TMM_SKIP), multi = there && !(M & TMM_SINGLE), rs = (M & TMM_RS) != 0;
        const uint32_t A = (uint32_t)__ballot(lane < 32 && multi), R = (uint32_t)__ballot(lane < 32 && rs);
        const uint32_t below = lane < 32 ? (1u << lane) - 1u : 0xffffffffu;
        const uint32_t rb = R & below;
        const uint32_t seg = rb ? below & ~((1u << (31 - __clz(rb))) - 1u) : below;
        const bool ob = (A & seg) != 0u || (!rb && open_in);
        if (rs && ob) M |= TMM_CLOSE;
        if (multi && (rs || !ob)) M |= TMM_FIRST;
        const uint32_t segl = R ? ~((1u << (31 - __clz(R))) - 1u) : 0xffffffffu;      // what the next group inherits
        open_in = ((A & segl) != 0u || (!R && open_in)) ? 1u : 0u;
        return M;
    };
    auto consume = [&](int g, const tm_u32x4 (&E)[TM_GROUP], uint32_t M0) {
        const uint32_t M = verdicts(M0);
#pragma unroll
        for (int k = 0; k < TM_GROUP; ++k) {
            if ((uint32_t)g * TM_GROUP + k >= nblk) break;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t m = rl(M, k * 8 + u);
                if (u & 1) tm_add<true>(st, m, E[k][u >> 1], thr, pkl0, one);
                else tm_add<false>(st, m, E[k][u >> 1], thr, pkl0, one);
            }
        }
    };
    issue(0, EA, MA);
    int g = 0;
    while (true) {
        if (g + 1 < ng) issue(g + 1, EB, MB);
        consume(g, EA, MA);
        if (++g >= ng) break;
        if (g + 1 < ng) issue(g + 1, EA, MA);
        consume(g, EB, MB);
        if (++g >= ng) break;
    }
    st.nc += st.mask & 0x10001u; st.mask = 0;                  // the run left open at the end (mask is 0 when none is)
}

// Workgroup = two waves = one job at a time: each wave walks half of the job's entries (cut at a run start) into the job's planes
// (8 KB per workgroup: 4 KB per wave, which is what lets 6-8 waves per SIMD be resident), then each wave finishes one cell type's unit.
__global__ __launch_bounds__(TMW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(8))) void k_tm_walk(CountArgs a, TmArgs tm) {
    __shared__ __attribute__((aligned(8192))) uint32_t planes[2][2][8 * 64];      // [cell type][plane][symbol row x lane]: the cell type is bit 12 of an address
    __shared__ uint32_t nc_sh[2][64];                                              // per cell type and lane: runs that counted an event
    __shared__ WaveBook books[TMW_WAVES];
    __shared__ uint32_t s_ck;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: says so to the compiler)
    uint32_t* pl = &planes[0][0][0];
    WaveBook& book = books[wv];
    book_init(book, lane);
    if (lane == 0) book.src = 1;
    const uint32_t thr = bq_threshold(a), pkl0 = lds_addr(pl + lane);
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));
    // a workgroup's first chunk is its own index, later ones come off the queue; a chunk = consecutive jobs of about TM_CHUNK_WORK work
    for (bool first = true;; first = false) {
        __syncthreads();
        if (threadIdx.x == 0) s_ck = first ? blockIdx.x : (uint32_t)atomicAdd(&a.scalars[SC_QSMALL], 1ull) + gridDim.x;
        __syncthreads();
        const uint32_t ck = rl(s_ck, 0);
        if (ck >= tm.nchunks) break;
        const uint32_t jx_end = rl(tm.chunk_start[ck + 1], 0);
        for (uint32_t jx = rl(tm.chunk_start[ck], 0); jx < jx_end; ++jx) {
            uint32_t jw = 0;
            if (lane < 8) jw = reinterpret_cast<const uint32_t*>(tm.jobs + jx)[lane];
            const uint32_t e0 = rl(jw, 0), e1 = rl(jw, 1), w0 = rl(jw, 2), slab = rl(jw, 3), nj = rl(jw, 4), tcnt = rl(jw, 5), tile = rl(jw, 6), emid = rl(jw, 7);
            if (tile < a.tile_lo || tile >= a.tile_hi) continue;
            const int2 geom = a.ne_geom[w0];
            const int tid = rl((uint32_t)geom.y, 0) & 0xffffff;
            const int32_t tstart = (int32_t)rl((uint32_t)geom.x, 0);
            int refb = 'N';
            if (nj == 1) { const int64_t pos = (int64_t)tstart + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            __syncthreads();                                   // both waves are done with the job before
#pragma unroll
            for (int i = 0; i < 2 * 2 * 8 * 64 / (4 * TMW_WAVES * 64); ++i) reinterpret_cast<uint4*>(pl)[i * (TMW_WAVES * 64) + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
            (&nc_sh[0][0])[threadIdx.x] = 0;
            __syncthreads();
            TmState st; st.nc = 0; st.mask = 0;
            const uint32_t s0 = wv ? emid : e0, s1 = wv ? e1 : emid;
            if (s1 > s0) {
                tm_walk_range(tm, st, s0, s1, thr, pkl0, one, lane);
                if (st.nc & 0xffffu) atomicAdd(&nc_sh[0][lane], st.nc & 0xffffu);
                if (st.nc >> 16) atomicAdd(&nc_sh[1][lane], st.nc >> 16);
            }
            __syncthreads();
            // the tile's units: wave = cell type
            if (wv < a.n_ct) {
                const int ct = wv;
                const uint32_t* pc = pl + ct * 1024;
                uint32_t dp = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp += pc[512 + k * 64 + lane] & 0xffffu;
                const TmCounters tot{pc, lane, dp - nc_sh[ct][lane]};
                if (nj == 1) {
                    if (tcnt <= 256u) emit_unit<TmCounters, true>(a, tot, w0 + ct, ct, tid, tstart, lane, &book, false, refb, 0);
                    else emit_unit<TmCounters, false>(a, tot, w0 + ct, ct, tid, tstart, lane, &book, false, refb, 1);
                } else {
                    uint32_t* dst = a.macc + (uint64_t)(slab + (uint32_t)ct * nj) * (NCTR * 64);
                    dst[lane] = tot.NCDUP();
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t lo = pc[k * 64 + lane], hi = pc[512 + k * 64 + lane];
                        dst[(1 + k) * 64 + lane] = hi >> 16; dst[(9 + k) * 64 + lane] = hi & 0xffffu;
                        dst[(17 + k) * 64 + lane] = lo & 0xfffffu; dst[(25 + k) * 64 + lane] = lo >> 20;
                    }
                }
            }
        }
    }
    lds_fence();
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t rt = 0, cols = 0, rsrc = 0;
        for (int w = 0; w < TMW_WAVES; ++w) {
            const WaveBook& b = books[w];
            if (lane < a.n_ct) rt += b.rows_true[lane];
            cols += b.cols; rsrc += b.rows_src;
        }
        if (lane < a.n_ct && rt) atomicAdd(&a.scalars[SC_ROWS + lane], (unsigned long long)rt);
        if (lane == 0) {
            if (cols) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)cols);
            if (rsrc) atomicAdd(&a.scalars[SC_ROWS_SRC + 1], (unsigned long long)rsrc);
        }
    }
}

static bool tm_key_matches(const lsg_ctx* c, const lsg_count_params* p) {
    return c->tm_valid && c->tm_key[0] == (int64_t)p->min_mq && c->tm_key[1] == (int64_t)p->flag_exclude && c->tm_key[2] == (int64_t)p->ignore_orphans &&
           c->tm_key[3] == (int64_t)c->n_ct;
}

static int build_tm(lsg_ctx* c, const lsg_count_params* p) {
    if (tm_key_matches(c, p)) return 0;
    c->tm_valid = false; c->tm_usable = false;
    if (build_index(c)) return -1;
    hipStream_t st = c->stream;
    const uint64_t N = c->ix_n;
    const uint32_t T = c->n_tiles;
    c->tm_key[0] = p->min_mq; c->tm_key[1] = p->flag_exclude; c->tm_key[2] = p->ignore_orphans; c->tm_key[3] = c->n_ct;
    if (N == 0 || N >= 0x7fffffffull) { c->tm_valid = true; return 0; }
    DevBuf &S = c->bt[6], &per_tile = c->bt[7], &offs = c->bt[8];
    auto done = [&](int rc) { return rc; };
    if (S.reserve((N + 2) * 4) || per_tile.reserve((size_t)(T + 2) * 4 * 6) || offs.reserve((size_t)(T + 2) * 4 * 5 + 64)) return done(-1);      // (offs: + two words behind the five arrays)
    TmAdm adm{c->d_ix2.as<uint32_t>(), p->flag_exclude, p->min_mq, p->ignore_orphans};
    {
        hipcub::CountingInputIterator<uint32_t> iota(0);
        hipcub::TransformInputIterator<uint32_t, TmAdm, hipcub::CountingInputIterator<uint32_t>> it(iota, adm);
        SCAN_U32(it, S.as<uint32_t>(), N + 1);
    }
    uint32_t* cnt = per_tile.as<uint32_t>(); uint32_t* blk = cnt + (T + 2); uint32_t* ne = blk + (T + 2); uint32_t* nj = ne + (T + 2);
    uint32_t* slabs = nj + (T + 2); uint32_t* multi = slabs + (T + 2);
    uint32_t* blk_off = offs.as<uint32_t>(); uint32_t* ne_off = blk_off + (T + 2); uint32_t* job_off = ne_off + (T + 2);
    uint32_t* slab_off = job_off + (T + 2); uint32_t* multi_off = slab_off + (T + 2);
    uint32_t* d_maxjob = multi_off + (T + 2);
    // jobs as long as the planes' fields allow (fewer slabs) — unless the load is small (one rank's share of a sharded job): then every
    // resident pair of waves should still get several
    uint32_t job_tgt = TM_JOB_TGT;
    { const uint64_t per = N / ((uint64_t)c->n_cus * 14 * 4); if (per < job_tgt) job_tgt = (uint32_t)(per < 768 ? 768 : per); }
    hipLaunchKernelGGL(k_tm_tiles, dim3((T + 256) / 256), dim3(256), 0, st, c->d_tile_off.as<uint32_t>(), S.as<uint32_t>(), T, c->n_ct, job_tgt, cnt, blk, ne, nj, slabs, multi);
    SCAN_U32(blk, blk_off, T + 1); SCAN_U32(ne, ne_off, T + 1); SCAN_U32(nj, job_off, T + 1); SCAN_U32(slabs, slab_off, T + 1); SCAN_U32(multi, multi_off, T + 1);
    LSG_HIP(hipMemsetAsync(d_maxjob, 0, 8, st));
    uint32_t tot[5] = {0, 0, 0, 0, 0};
    uint32_t* srcs[5] = {blk_off, ne_off, job_off, slab_off, multi_off};
    for (int i = 0; i < 5; ++i) LSG_HIP(hipMemcpyAsync(&tot[i], srcs[i] + T, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    const uint32_t nblk = tot[0], n_net = tot[1], njobs = tot[2], n_slabs = tot[3], n_mt = tot[4];
    const uint64_t np = (uint64_t)nblk * 8;
    c->tm_np = np; c->tm_nblk = nblk; c->tm_njobs = njobs; c->tm_n_ne = n_net * (uint32_t)c->n_ct; c->tm_n_multi = n_mt * (uint32_t)c->n_ct; c->tm_n_slabs = n_slabs;
    if (nblk == 0) { c->tm_valid = true; return done(0); }
    const size_t n_ne = c->tm_n_ne;
    if (c->tm[TM_STORE].reserve(((size_t)nblk + TM_GROUP) * 1024) || c->tm[TM_S0].reserve((np + 16) * 4) || c->tm[TM_B].reserve(np + 16) ||
        c->tm[TM_LINE].reserve((np + 16) * 4) || c->tm[TM_META].reserve((np + 8 * (TM_GROUP + 1)) * 4) || c->tm[TM_BLK_TILE].reserve(((size_t)nblk + 2) * 4) ||
        c->tm[TM_JOBS].reserve(((size_t)njobs + 1) * sizeof(TmJob)) || c->tm[TM_EXT].reserve(((size_t)nblk + TM_GROUP + 2) * 2) || c->tm[TM_NE_UNITS].reserve((n_ne + 2) * 4) || c->tm[TM_NE_GEOM].reserve((n_ne + 2) * 8) ||
        c->tm[TM_NE_NSLOT].reserve((n_ne + 2) * 4) || c->tm[TM_NE_ACC].reserve((n_ne + 2) * 4) || c->tm[TM_MULTI].reserve(((size_t)c->tm_n_multi + 2) * 4))
        return done(-1);
    uint32_t* s0 = c->tm[TM_S0].as<uint32_t>(); uint32_t* line = c->tm[TM_LINE].as<uint32_t>(); uint8_t* b8 = c->tm[TM_B].as<uint8_t>();
    uint32_t* blk_tile = c->tm[TM_BLK_TILE].as<uint32_t>();
    LSG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s0), (int)TM_PAD_S0, np + 16, st));
    LSG_HIP(hipMemsetAsync(b8, 0, np + 16, st));
    LSG_HIP(hipMemsetAsync(line, 0, (np + 16) * 4, st));
    LSG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->tm[TM_META].p), (int)TMM_SKIP, np + 8 * (TM_GROUP + 1), st));
    LSG_HIP(hipMemsetAsync(c->tm[TM_NE_NSLOT].p, 0, (n_ne + 2) * 4, st));
    LSG_HIP(hipMemsetAsync(c->tm[TM_NE_ACC].p, 0, (n_ne + 2) * 4, st));
    hipLaunchKernelGGL(k_tm_blk_tile, dim3((nblk + 255) / 256), dim3(256), 0, st, blk_off, T, nblk, blk_tile);
    hipLaunchKernelGGL(k_tm_fill, dim3((unsigned)(c->n_cus * 16)), dim3(256), 0, st, adm, c->d_ix0.as<uint32_t>(), c->d_ix1.as<uint32_t>(), c->d_ix2.as<uint32_t>(), N,
                       c->d_tile_off.as<uint32_t>(), T, S.as<uint32_t>(), blk_off, s0, line, b8);
    hipLaunchKernelGGL(k_tm_runs, dim3((unsigned)(c->n_cus * 16)), dim3(256), 0, st, s0, b8, np, blk_off, blk_tile);
    hipLaunchKernelGGL(k_tm_gather, dim3((unsigned)((((uint64_t)nblk + TMG_BLOCKS - 1) / TMG_BLOCKS * 64 + 255) / 256)), dim3(256), 0, st, c->rd.events, s0, line, nblk, c->tm[TM_STORE].as<uint4>(), c->tm[TM_EXT].as<uint16_t>());
    {
        CountArgs a{};
        a.tile_base = c->d_tile_base.as<uint32_t>(); a.n_contigs = c->n_contigs; a.n_ct = c->n_ct;
        hipLaunchKernelGGL(k_tm_jobs, dim3((T + 255) / 256), dim3(256), 0, st, a, s0, cnt, blk_off, ne_off, nj, job_off, slab_off, multi_off, T,
                           c->tm[TM_JOBS].as<TmJob>(), c->tm[TM_NE_UNITS].as<uint32_t>(), c->tm[TM_NE_GEOM].as<int2>(), c->tm[TM_NE_NSLOT].as<uint32_t>(),
                           c->tm[TM_NE_ACC].as<uint32_t>(), c->tm[TM_MULTI].as<uint32_t>(), d_maxjob);
    }
    uint32_t n_chunks = 0;
    {   // static work-balanced chunks of the job list
        DevBuf& pex = c->bt[9];
        // every workgroup of the walk should get several chunks: a small load (one rank's share of a sharded job) is cut finer
        const uint64_t total_work = (uint64_t)np + (uint64_t)njobs * TM_JOB_W0;
        uint64_t cw = total_work / ((uint64_t)c->n_cus * 14 * 6);
        const uint32_t chunk_work = (uint32_t)(cw < 256 ? 256 : (cw > TM_CHUNK_WORK ? TM_CHUNK_WORK : cw));
        if (pex.reserve(((size_t)njobs + 2) * 4) || c->tm[TM_CHUNKS].reserve(((size_t)(total_work / chunk_work) + 4) * 4)) { return done(-1); }
        hipcub::CountingInputIterator<uint32_t> iota(0);
        TmJobWork wf{c->tm[TM_JOBS].as<TmJob>()};
        hipcub::TransformInputIterator<uint32_t, TmJobWork, hipcub::CountingInputIterator<uint32_t>> it(iota, wf);
        size_t tb_ = 0;
        hipError_t e1 = hipcub::DeviceScan::ExclusiveSum(nullptr, tb_, it, pex.as<uint32_t>(), (int)njobs, st);
        if (e1 != hipSuccess || cub_tmp(c, tb_)) { return done(-1); }
        tb_ = c->d_cub_tmp.cap;
        e1 = hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb_, it, pex.as<uint32_t>(), (int)njobs, st);
        hipLaunchKernelGGL(k_tm_chunks, dim3((njobs + 255) / 256), dim3(256), 0, st, pex.as<uint32_t>(), njobs, chunk_work, c->tm[TM_CHUNKS].as<uint32_t>(), d_maxjob + 1);
        hipError_t e2 = hipMemcpyAsync(&n_chunks, d_maxjob + 1, 4, hipMemcpyDeviceToHost, st);
        hipError_t e3 = hipStreamSynchronize(st);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { set_error("lsg_pileup_count: tile-major chunk table failed"); return done(-1); }
    }
    c->tm_nchunks = n_chunks;
    uint32_t max_job = 0;
    LSG_HIP(hipMemcpyAsync(&max_job, d_maxjob, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipGetLastError());
    LSG_HIP(hipStreamSynchronize(st));
    c->tm_usable = max_job == 0;                  // a job a single barcode's run stretched past the planes' fields: the index path counts this load
    c->tm_valid = true;
    return done(0);
}

static int run_count_tm(lsg_ctx* c, const lsg_count_params* p) {
    hipStream_t st = c->stream;
    const uint32_t n_ne = c->tm_n_ne;
    const int64_t R = c->rd.n_reads;
    if (c->d_read_key.reserve((size_t)(R + 1) * 4) || c->d_scalars.reserve(SC_COUNT * 8) || c->d_ne_units.reserve(((size_t)n_ne + 2) * 4) ||
        c->d_ne_mask.reserve(((size_t)n_ne + 2) * 8) || c->d_ne_rowbase.reserve(((size_t)n_ne + 2) * 4) || c->ws[WS_NE_NSLOT].reserve(((size_t)n_ne + 2) * 4) ||
        c->ws[WS_NE_ACC].reserve(((size_t)n_ne + 2) * 4) || c->ws[WS_NE_GEOM].reserve(((size_t)n_ne + 2) * 8) || c->ws[WS_MULTI_LIST].reserve(((size_t)c->tm_n_multi + 2) * 4) ||
        c->ws[WS_MACC].reserve(((size_t)c->tm_n_slabs + 1) * NCTR * 64 * 4) || c->d_ix_stat.reserve(IX_STAT_SLOTS * 64))
        return -1;
    const unsigned grid_walk = (unsigned)(c->n_cus * tune_int("LSG_GRID_TM", 14));
    const unsigned grid_fin = (unsigned)(c->n_cus * 8);
    {   // row buffers: bound + one open arena per emitting wave and format
        uint64_t want_rows = (uint64_t)n_ne * TILE_W;
        if (p->min_dp > 0) {
            const uint64_t by_depth = (uint64_t)c->rd.n_events / (uint64_t)p->min_dp + 64;
            if (by_depth < want_rows) want_rows = by_depth;
        }
        const uint64_t emitters = (uint64_t)grid_walk * TMW_WAVES * 2 + grid_fin;
        uint64_t arena = want_rows / (emitters * 8) / ARENA * ARENA;
        c->arena = (uint32_t)(arena < (uint64_t)ARENA ? (uint64_t)ARENA : (arena > 8ull * ARENA ? 8ull * ARENA : arena));
        want_rows += emitters * c->arena + 64;
        want_rows = (want_rows + 63) / 64 * 64 + 64;
        if (want_rows > c->row_cap) c->row_cap = want_rows;
        for (int i = 0; i < c->n_ct; ++i)
            if (c->d_rows[i].reserve((size_t)c->row_cap * ROW_STORED_WORDS * 4)) return -1;
    }
    c->n_ne = n_ne; c->n_slots = 0; c->n_multi = c->tm_n_multi;
    CountArgs a{};
    fill_args(c, p, a);
    TmArgs tm{};
    tm.store = c->tm[TM_STORE].as<uint4>(); tm.s0 = c->tm[TM_S0].as<uint32_t>(); tm.b = c->tm[TM_B].as<uint8_t>(); tm.meta = c->tm[TM_META].as<uint32_t>();
    tm.blk_tile = c->tm[TM_BLK_TILE].as<uint32_t>(); tm.jobs = c->tm[TM_JOBS].as<TmJob>(); tm.np = c->tm_np; tm.nblk = c->tm_nblk; tm.njobs = c->tm_njobs; tm.nchunks = c->tm_nchunks; tm.chunk_start = c->tm[TM_CHUNKS].as<uint32_t>(); tm.ext = c->tm[TM_EXT].as<uint16_t>();
    LSG_HIP(hipEventRecord(c->ev[0], st));
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, st));
    LSG_HIP(hipMemsetAsync(c->d_ix_stat.p, 0, IX_STAT_SLOTS * 64, st));
    if (n_ne) {
        // the static unit tables in the places the call stage and the exports read
        LSG_HIP(hipMemcpyAsync(c->d_ne_units.p, c->tm[TM_NE_UNITS].p, (size_t)n_ne * 4, hipMemcpyDeviceToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->ws[WS_NE_GEOM].p, c->tm[TM_NE_GEOM].p, (size_t)n_ne * 8, hipMemcpyDeviceToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->ws[WS_NE_NSLOT].p, c->tm[TM_NE_NSLOT].p, ((size_t)n_ne + 1) * 4, hipMemcpyDeviceToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->ws[WS_NE_ACC].p, c->tm[TM_NE_ACC].p, ((size_t)n_ne + 1) * 4, hipMemcpyDeviceToDevice, st));
        if (c->tm_n_multi) LSG_HIP(hipMemcpyAsync(c->ws[WS_MULTI_LIST].p, c->tm[TM_MULTI].p, (size_t)c->tm_n_multi * 4, hipMemcpyDeviceToDevice, st));
        LSG_HIP(hipMemsetAsync(c->d_ne_mask.p, 0, ((size_t)n_ne + 1) * 8, st));
        LSG_HIP(hipMemsetAsync(c->d_ne_rowbase.p, 0, ((size_t)n_ne + 1) * 4, st));
    }
    if (R > 0) { unsigned g = (unsigned)((R + 255) / 256); if (g > (unsigned)(c->n_cus * 8)) g = (unsigned)(c->n_cus * 8); hipLaunchKernelGGL(k_read_key, dim3(g), dim3(256), 0, st, a); }
    if (c->tm_nblk) {
        hipLaunchKernelGGL(k_tm_resolve, dim3((c->tm_nblk + 255) / 256), dim3(256), 0, st, a, tm, c->d_ix_stat.as<unsigned long long>());
        hipLaunchKernelGGL(k_resolve_stats, dim3(1), dim3(256), 0, st, a, c->d_ix_stat.as<unsigned long long>());
    }
    LSG_HIP(hipEventRecord(c->ev[1], st));
    LSG_HIP(hipEventRecord(c->ev[2], st));
    LSG_HIP(hipEventRecord(c->ev[3], st));
    if (c->tm_njobs) hipLaunchKernelGGL(k_tm_walk, dim3(grid_walk), dim3(TMW_WAVES * 64), 0, st, a, tm);
    LSG_HIP(hipEventRecord(c->ev[4], st));
    if (c->tm_n_multi) hipLaunchKernelGGL(k_finalize_multi, dim3(c->tm_n_multi < grid_fin ? c->tm_n_multi : grid_fin), dim3(FIN_THREADS), 0, st, a);
    LSG_HIP(hipEventRecord(c->ev[5], st));
    LSG_HIP(hipGetLastError());
    unsigned long long sc[SC_COUNT];
    if (read_scalars(c, sc)) return -1;
    if (sc[SC_OVERFLOW]) { set_error("lsg_pileup_count: row buffer overflow (cap %llu)", (unsigned long long)c->row_cap); return -3; }
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->n_rows[i] = (int64_t)sc[SC_ROWS + i];
    c->n_columns = (int64_t)sc[SC_COLS];
    c->stats.n_reads_admitted = (int64_t)sc[SC_READS];
    c->stats.n_segs_admitted = (int64_t)sc[SC_SEGS];
    c->stats.n_events_admitted = (int64_t)sc[SC_EVENTS];
    c->stats.n_units = n_ne;
    c->stats.n_deep_units = c->tm_n_multi;
    c->stats.n_entries = (int64_t)sc[SC_NENT];
    float ms = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->stats.ms_bin = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[1], c->ev[5])); c->stats.ms_deep = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); c->stats.ms_walk = ms;
    c->stats.ms_wave = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[5])); c->stats.ms_total = ms;
    for (int i = 0; i < 4; ++i) { c->stats.rows_by_kernel[i] = (int64_t)sc[SC_ROWS_SRC + i]; c->stats.events_by_kernel[i] = 0; }
    c->stats.events_by_kernel[1] = (int64_t)sc[SC_EVENTS];           // one kernel reads every event
    c->stats.n_events_wave = 0; c->stats.n_events_deep = (int64_t)sc[SC_EVENTS];
    c->stats.n_rows_deep = (int64_t)sc[SC_ROWS_DEEP];
    c->stats.n_rows_wave = -c->stats.n_rows_deep;
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->stats.n_rows_wave += c->n_rows[i];
    c->last_params = *p;
    c->counted = true;
    c->called = false;
    return 0;
}

// 0 auto (a load counted LAYOUT_AUTO_AFTER times under the same read filters gets them at the next count), 1 eager, 2 never;
// LSG_LAYOUT=auto|eager|never overrides what lsg_set_layout_policy left.  Building costs about as much as ten counts save (C2: 60 ms
// against 6 ms per count), so auto waits for evidence that the load is being counted over and over; callers that know say so
// (lsg_prepare_counts).
constexpr uint32_t LAYOUT_AUTO_AFTER = 3;
static int layout_policy(const lsg_ctx* c) {
    const char* e = getenv("LSG_LAYOUT");
    if (e && *e) return e[0] == 'e' ? 1 : (e[0] == 'n' ? 2 : 0);
    return c->layout_policy;
}
// builds the tile index and the tile-major store for these read filters when they can serve the counts (rc != 0: a real failure)
int prepare_layout(lsg_ctx* c, const lsg_count_params* p) {
    hipStream_t st = c->stream;
    if (!(c->n_ct <= 2 && c->n_ct > 0 && c->rd.n_reads < 0x7fffffffll) || getenv("LSG_COUNT_PASS") || getenv("LSG_NO_INDEX") || getenv("LSG_NO_TM") || layout_policy(c) == 2) return 0;
    if (tile_capacities(c)) return -1;
    if (depth_cap_drops(c, p)) return -1;
    if (c->has_drops || tm_key_matches(c, p)) return 0;
    // index + store + the build's temporaries take ~200 bytes of device memory per entry: a load that leaves less free
    // is counted without them (the scatter path needs ~50), and so is one whose build fails half-way
    size_t mem_free = 0, mem_total = 0;
    (void)hipMemGetInfo(&mem_free, &mem_total);
    size_t held = 0;
    for (auto& b : c->tm) held += b.cap;
    const char* bpe_env = getenv("LSG_TM_BYTES_PER_ENTRY");          // (tests: pretend the store is larger than it is)
    const uint64_t bpe = bpe_env && *bpe_env ? strtoull(bpe_env, nullptr, 10) : 200ull;
    const bool fits = (uint64_t)c->entries_upper * bpe < (uint64_t)mem_free + held;
    const auto t0 = std::chrono::steady_clock::now();
    if (!fits || build_tm(c, p)) {
        (void)hipGetLastError();
        for (auto& b : c->tm) b.release();
        c->tm_key[0] = p->min_mq; c->tm_key[1] = p->flag_exclude; c->tm_key[2] = p->ignore_orphans; c->tm_key[3] = c->n_ct;
        c->tm_valid = true; c->tm_usable = false;          // (until the reads or the filters change)
        if (getenv("LSG_TIMING")) fprintf(stderr, "[lsg] tile-major store not built (%s): counting without it\n", fits ? "build failed" : "device memory");
    }
    LSG_HIP(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->layout_build_ms += ms;
    if (getenv("LSG_TIMING") && c->tm_usable)
        fprintf(stderr, "[lsg] tile index + tile-major store of %llu entries (%u blocks, %u jobs) built in %.2f ms\n", (unsigned long long)c->ix_n, c->tm_nblk, c->tm_njobs, ms);
    return 0;
}

int run_count(lsg_ctx* c, const lsg_count_params* p) {
    if (c->n_contigs <= 0) { set_error("lsg_pileup_count: no contigs set"); return -2; }
    if (c->n_ct <= 0) { set_error("lsg_pileup_count: no barcodes set"); return -2; }
    if (!c->rd.events && c->rd.n_events > 0) { set_error("lsg_pileup_count: no reads loaded"); return -2; }
    if ((((uint64_t)(uintptr_t)(c->rd.events + c->rd.n_events) + 256) >> 47) != 0) {      // entry meta keeps address bits 32..46
        set_error("lsg_pileup_count: resident events lie above the 47-bit address range"); return -1;
    }
    for (int t = 0; t < c->n_contigs; ++t)
        if (!c->ref_ptr[t]) { set_error("lsg_pileup_count: reference of contig %d not loaded", t); return -2; }
    hipStream_t st = c->stream;
    // Per-load structures (tile index, tile-major store) serve <= 2 cell types and counts without depth-cap drops.  They cost more to
    // build than several counts on the scatter path, so (policy auto) the first LAYOUT_AUTO_AFTER counts of a load under given read
    // filters run without them and the next one builds them; lsg_prepare_counts or policy eager build at once, never leaves them out.
    c->tm_path = false;
    bool want_layout = false, want_store = false;
    if (c->n_ct <= 2 && c->rd.n_reads < 0x7fffffffll && !getenv("LSG_COUNT_PASS") && !getenv("LSG_NO_INDEX")) {
        const bool same_key = c->seen_key[0] == (int64_t)p->min_mq && c->seen_key[1] == (int64_t)p->flag_exclude && c->seen_key[2] == (int64_t)p->ignore_orphans &&
                              c->seen_key[3] == (int64_t)c->n_ct;
        if (!same_key) { c->seen_key[0] = p->min_mq; c->seen_key[1] = p->flag_exclude; c->seen_key[2] = p->ignore_orphans; c->seen_key[3] = c->n_ct; c->seen_counts = 0; }
        const int policy = layout_policy(c);
        want_store = policy == 1 || (policy == 0 && (c->seen_counts >= LAYOUT_AUTO_AFTER || tm_key_matches(c, p)));
        want_layout = want_store || (policy == 0 && c->index_valid);      // an index that exists already serves other filters too
        ++c->seen_counts;
    }
    if (want_store && !getenv("LSG_NO_TM")) {
        if (int rc = prepare_layout(c, p)) return rc;
        if (!c->has_drops && c->tm_usable) { c->tm_path = true; c->index_path = false; return run_count_tm(c, p); }
    }
    const uint32_t n_units = c->n_tiles * (uint32_t)c->n_ct;
    const int64_t R = c->rd.n_reads, S = c->rd.n_segs;
    uint32_t max_ct = 1; for (int i = 0; i < c->n_ct; ++i) max_ct = c->ct_size[i] > max_ct ? c->ct_size[i] : max_ct;
    const uint64_t EU = c->entries_upper;
    const uint64_t ne_cap = n_units < EU ? n_units : EU;
    const uint64_t slot_cap = ne_cap + EU / SUBT + 16;

    if (c->d_read_key.reserve((size_t)(R + 1) * 4) || c->d_unit_cnt.reserve(((size_t)n_units + 2) * 4) ||
        c->d_unit_off.reserve(((size_t)n_units + 2) * 4) || c->d_unit_fill.reserve(((size_t)n_units + 2) * 4) ||
        c->d_scalars.reserve(SC_COUNT * 8) || c->d_ne_units.reserve((ne_cap + 2) * 4) || c->d_ne_mask.reserve((ne_cap + 2) * 8) ||
        c->d_ne_rowbase.reserve((ne_cap + 2) * 4) || c->ws[WS_NE_NSLOT].reserve((ne_cap + 2) * 4) ||
        c->ws[WS_NE_SLOT_BASE].reserve((ne_cap + 2) * 4) || c->ws[WS_NE_ACC].reserve((ne_cap + 2) * 4) ||
        c->ws[WS_NE_GEOM].reserve((ne_cap + 2) * 8) || c->ws[WS_SLOT_W].reserve((slot_cap + 2) * 4) ||
        c->ws[WS_SLOT_CNT].reserve((slot_cap + 2) * 4) || c->ws[WS_SLOT_OFF].reserve((slot_cap + 2) * 4) ||
        c->ws[WS_SLOT_LIST].reserve((slot_cap + 2) * 4) ||
        c->ws[WS_MULTI_LIST].reserve((EU / CAPB + 16) * 4) || c->ws[WS_ENT].reserve((EU + 1) * 16 + 64) || c->ws[WS_SEG_INFO].reserve(((size_t)S + 1) * 8) || c->ws[WS_REC].reserve((EU + 1) * 16 + 256) ||
        c->ws[WS_SLICES].reserve((slot_cap + 2) * (NSLICE + 1) * 4) || c->ws[WS_SLOT_PEX].reserve((slot_cap + 2) * 4) ||
        c->ws[WS_CHUNK_START].reserve(((EU + WORK_W0 * slot_cap) / CHUNK_EMIN + 4) * 4) || c->ws[WS_HUGE_LIST].reserve((EU / CAPB + 16) * 8))
        return -1;

    if (tile_capacities(c)) return -1;
    if (depth_cap_drops(c, p)) return -1;           // free unless some cell type's pileup buffer can reach max_depth (cached bound)
    // the tile index serves <= 2 cell types and counts without depth-cap drops (those are per read and rare: the scatter path takes them)
    c->index_path = want_layout && !c->has_drops;
    if (c->index_path && !c->index_valid) {
        size_t mem_free = 0, mem_total = 0;
        (void)hipMemGetInfo(&mem_free, &mem_total);
        if ((uint64_t)c->entries_upper * 64ull >= (uint64_t)mem_free) c->index_path = false;      // ~12 bytes per entry to keep, ~50 while it is sorted
    }
    if (c->index_path && !c->index_valid) {
        const auto t0 = std::chrono::steady_clock::now();
        if (build_index(c)) return -1;
        LSG_HIP(hipStreamSynchronize(c->stream));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        c->layout_build_ms += ms;
        if (getenv("LSG_TIMING")) fprintf(stderr, "[lsg] tile index of %llu entries built in %.2f ms\n", (unsigned long long)c->ix_n, ms);
    }
    if (c->index_path && (c->ix_n == 0 || c->d_ix_stat.reserve(IX_STAT_SLOTS * 64))) c->index_path = false;
    LSG_HIP(hipEventRecord(c->ev[0], st));
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, st));
    // only the units of the counted region (lsg_set_region) are ever touched
    const uint32_t u_lo = c->tile_lo * (uint32_t)c->n_ct;
    const uint32_t t_hi = c->tile_hi < c->n_tiles ? c->tile_hi : c->n_tiles;
    const uint32_t u_hi = t_hi * (uint32_t)c->n_ct;
    const uint32_t n_range = u_hi > u_lo ? u_hi - u_lo : 0;
    const uint32_t n_trange = t_hi > c->tile_lo ? t_hi - c->tile_lo : 0;
    c->n_ne = c->n_slots = c->n_multi = 0;
    CountArgs a{};
    fill_args(c, p, a);
    const bool two_ended = a.two_ended != 0;
    unsigned seg_grid = (unsigned)((S + 256 * BIN_SUPER - 1) / (256 * BIN_SUPER));
    if (seg_grid > (unsigned)(c->n_cus * 8)) seg_grid = (unsigned)(c->n_cus * 8);
    if (R > 0) { unsigned g = (unsigned)((R + 255) / 256); if (g > (unsigned)(c->n_cus * 8)) g = (unsigned)(c->n_cus * 8); hipLaunchKernelGGL(k_read_key, dim3(g), dim3(256), 0, st, a); }
    if (S > 0 && !a.inline_seg_info && !a.index_path) { unsigned g = (unsigned)((S + 255) / 256); if (g > (unsigned)(c->n_cus * 16)) g = (unsigned)(c->n_cus * 16); hipLaunchKernelGGL(k_seg_info, dim3(g), dim3(256), 0, st, a); }
    if (a.index_path) {
        // ONE streaming pass over the tile index: admission, cell type, order-preserving compaction into the walk's records
        const uint64_t n_chunks = (c->ix_n + IX_CHUNK - 1) / IX_CHUNK;
        if (n_range) LSG_HIP(hipMemsetAsync(c->d_unit_cnt.as<uint32_t>() + u_lo, 0, ((size_t)n_range + 1) * 4, st));
        LSG_HIP(hipMemsetAsync(c->d_ix_stat.p, 0, IX_STAT_SLOTS * 64, st));
        hipLaunchKernelGGL(k_resolve_agg, dim3((unsigned)n_chunks), dim3(RES_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)n_chunks), dim3(RES_THREADS), 0, st, a, c->d_ix_stat.as<unsigned long long>());
        hipLaunchKernelGGL(k_resolve_stats, dim3(1), dim3(256), 0, st, a, c->d_ix_stat.as<unsigned long long>());
    } else if (two_ended) {
        // ONE pass over the segments: every tile owns a static region of the entry buffer (tile_capacities), cell type 0 fills it
        // from the front and cell type 1 from the back, the units' sizes fall out of the cursors
        if (n_trange) {
            LSG_HIP(hipMemcpyAsync(a.cur_lo + c->tile_lo, a.tile_off + c->tile_lo, (size_t)n_trange * 4, hipMemcpyDeviceToDevice, st));
            LSG_HIP(hipMemcpyAsync(a.cur_hi + c->tile_lo, a.tile_off + c->tile_lo + 1, (size_t)n_trange * 4, hipMemcpyDeviceToDevice, st));
            if (S > 0) hipLaunchKernelGGL(k_bin_segments<2>, dim3(seg_grid), dim3(256), 0, st, a);
            { unsigned g = (n_trange + 255) / 256; if (g > (unsigned)(c->n_cus * 8)) g = (unsigned)(c->n_cus * 8); hipLaunchKernelGGL(k_units_from_cursors, dim3(g), dim3(256), 0, st, a); }
        }
    } else {
        LSG_HIP(hipMemsetAsync(c->d_unit_cnt.as<uint32_t>() + u_lo, 0, ((size_t)n_range + 1) * 4, st));
        if (S > 0) hipLaunchKernelGGL(k_bin_segments<0>, dim3(seg_grid), dim3(256), 0, st, a);
        SCAN_U32(a.unit_cnt + u_lo, a.unit_off + u_lo, n_range + 1);          // entry regions of buffer A, unit by unit
    }

    // non-empty units, in genomic order
    hipcub::CountingInputIterator<uint32_t> cnt_it(0), unit_it(u_lo);
    uint32_t* d_nne = reinterpret_cast<uint32_t*>(a.scalars + SC_NNE);
    {
        NonEmpty pred{a.unit_cnt};
        size_t tb = 0;
        LSG_HIP(hipcub::DeviceSelect::If(nullptr, tb, unit_it, a.ne_units, d_nne, (int)n_range, pred, st));
        if (cub_tmp(c, tb)) return -1;
        tb = c->d_cub_tmp.cap;
        LSG_HIP(hipcub::DeviceSelect::If(c->d_cub_tmp.p, tb, unit_it, a.ne_units, d_nne, (int)n_range, pred, st));
    }
    unsigned long long sc[SC_COUNT];
    uint32_t* pin32 = reinterpret_cast<uint32_t*>(c->h_pin + SC_COUNT + 8);   // small reads that ride on read_scalars' synchronisation
    if (!two_ended && !a.index_path) LSG_HIP(hipMemcpyAsync(pin32, c->d_unit_off.as<uint32_t>() + u_hi, 4, hipMemcpyDeviceToHost, st));
    if (read_scalars(c, sc)) return -1;
    if (sc[SC_OVERFLOW]) { set_error("lsg_pileup_count: a tile got more entries than its static capacity"); return -1; }
    const uint32_t total_entries = (two_ended || a.index_path) ? (uint32_t)sc[SC_NENT] : pin32[0];                  // statistics
    const uint32_t n_ne = (uint32_t)(sc[SC_NNE] & 0xffffffffull);
    c->n_ne = n_ne;
    fill_args(c, p, a);

    // launch-shape knobs
    const unsigned grid_block = (unsigned)(c->n_cus * 2);      // k_pileup_huge
    const unsigned grid_walk = (unsigned)(c->n_cus * tune_int("LSG_GRID_WALK", 8));       // k_walk_block
    const unsigned grid_wave = (unsigned)(c->n_cus * tune_int("LSG_GRID_WAVE", c->index_path ? 8 : 4));

    if (n_ne > 0) {
        // slot plan: deep units are cut into barcode-range slots
        hipLaunchKernelGGL(k_unit_plan, dim3((n_ne + 1 + 255) / 256), dim3(256), 0, st, a);
        SCAN_U32(a.ne_nslot, a.ne_slot_base, n_ne + 1);
        SCAN_U32(a.ne_acc, a.ne_acc, n_ne + 1);
        LSG_HIP(hipMemcpyAsync(pin32 + 2, a.ne_acc + n_ne, 4, hipMemcpyDeviceToHost, st));
        LSG_HIP(hipMemcpyAsync(pin32 + 4, a.ne_slot_base + n_ne, 4, hipMemcpyDeviceToHost, st));
        if (read_scalars(c, sc)) return -1;
        const uint32_t n_slabs = pin32[2];
        c->n_slots = pin32[4];
        c->n_multi = (uint32_t)sc[SC_NMULTI];
        if (c->n_slots > slot_cap) { set_error("lsg_pileup_count: slot plan exceeds its bound"); return -1; }
        if (c->ws[WS_MACC].reserve(((size_t)n_slabs + 1) * NCTR * 64 * 4)) return -1;
    }
    // row buffers: bound + arena slack
    {
        uint64_t want_rows = (uint64_t)n_ne * TILE_W;
        if (p->min_dp > 0) {
            uint64_t by_depth = (uint64_t)c->rd.n_events / (uint64_t)p->min_dp + 64;
            if (by_depth < want_rows) want_rows = by_depth;
        }
        // rows a wave reserves per allocation: large enough that the allocator words see few atomics (each takes ~90 per microsecond),
        // small enough that the open arenas stay a fraction of the rows themselves
        const uint64_t emitters = (uint64_t)grid_block + grid_walk + (uint64_t)grid_wave * WAVES_PER_BLOCK + (unsigned)(c->n_cus * 8);
        uint64_t arena = want_rows / (emitters * 8) / ARENA * ARENA;
        c->arena = (uint32_t)(arena < (uint64_t)ARENA ? (uint64_t)ARENA : (arena > 8ull * ARENA ? 8ull * ARENA : arena));
        want_rows += emitters * c->arena + 64;      // one open arena per emitting wave
        want_rows = (want_rows + 63) / 64 * 64 + 64;        // whole 64-row blocks (lsg::row_word), one spare: a unit's descriptor spans two
        // every cell type of THIS run needs planes of the current stride (a run with more cell types than any before it
        // finds row_cap large enough but its new buffers still empty)
        if (want_rows > c->row_cap) c->row_cap = want_rows;
        for (int i = 0; i < c->n_ct; ++i)
            if (c->d_rows[i].reserve((size_t)c->row_cap * ROW_STORED_WORDS * 4)) return -1;
    }
    fill_args(c, p, a);
    if (n_ne > 0) {
        hipLaunchKernelGGL(k_slot_init, dim3((n_ne + 255) / 256), dim3(256), 0, st, a);
        if (!two_ended && !a.index_path) {
            LSG_HIP(hipMemcpyAsync(a.unit_cursor + u_lo, a.unit_off + u_lo, ((size_t)n_range + 1) * 4, hipMemcpyDeviceToDevice, st));
            if (S > 0) hipLaunchKernelGGL(k_bin_segments<2>, dim3(seg_grid), dim3(256), 0, st, a);
        }
        // ---- work lists: small slots first, the rest reversed at the end
        {
            SmallSlot pred{a.slot_cnt, a.slot_w, a.ne_nslot};
            uint32_t* d_nsmall = reinterpret_cast<uint32_t*>(a.scalars + SC_NSMALL);
            size_t tb = 0;
            LSG_HIP(hipcub::DevicePartition::If(nullptr, tb, cnt_it, a.slot_list, d_nsmall, (int)c->n_slots, pred, st));
            if (cub_tmp(c, tb)) return -1;
            tb = c->d_cub_tmp.cap;
            LSG_HIP(hipcub::DevicePartition::If(c->d_cub_tmp.p, tb, cnt_it, a.slot_list, d_nsmall, (int)c->n_slots, pred, st));
        }
        {
            SlotWork wf{a.slot_list, a.slot_cnt, a.scalars};
            hipcub::TransformInputIterator<uint32_t, SlotWork, hipcub::CountingInputIterator<uint32_t>> work_it(cnt_it, wf);
            SCAN_U32(work_it, a.slot_pex, c->n_slots + 1);
            hipLaunchKernelGGL(k_chunk_starts, dim3((c->n_slots + 255) / 256), dim3(256), 0, st, a, (uint32_t)(grid_wave * WAVES_PER_BLOCK));
        }
        // small units (one wavefront each).  Measured (round 2): running this kernel on a second stream beside the deep units' sort and
        // grouping buys nothing — its persistent workgroups hold 128 KB of LDS and half the wave slots of every CU, k_sort_deep's
        // 1024-thread workgroups the other half, and whichever starts first starves the other — so everything stays on one stream.
        LSG_HIP(hipEventRecord(c->ev[2], st));
        static_assert(WIX_WAVES == WAVES_PER_BLOCK, "one grid size for both wave kernels");
        if (a.index_path) hipLaunchKernelGGL(k_wave_ix, dim3(grid_wave), dim3(WIX_WAVES * 64), 0, st, a);
        else hipLaunchKernelGGL(k_pileup_wave, dim3(grid_wave), dim3(WAVES_PER_BLOCK * 64), 0, st, a);
        LSG_HIP(hipEventRecord(c->ev[3], st));
        if (c->n_multi > 0) {
            MultiUnit pred{a.ne_nslot};
            uint32_t* d_nm = reinterpret_cast<uint32_t*>(a.scalars + SC_NMULTI_SEL);
            size_t tb = 0;
            LSG_HIP(hipcub::DeviceSelect::If(nullptr, tb, cnt_it, a.multi_list, d_nm, (int)n_ne, pred, st));
            if (cub_tmp(c, tb)) return -1;
            tb = c->d_cub_tmp.cap;
            LSG_HIP(hipcub::DeviceSelect::If(c->d_cub_tmp.p, tb, cnt_it, a.multi_list, d_nm, (int)n_ne, pred, st));
            unsigned sg = c->n_multi < (unsigned)(c->n_cus * 4) ? c->n_multi : (unsigned)(c->n_cus * 4);
            if (a.index_path) {
                // nothing to sort: the records are grouped already
            } else if (a.presorted) {
                const uint32_t r_cap = (max_ct + 63u) & ~63u;
                const size_t lds = ((size_t)2 * r_cap + MAXSUB + 1 + SORT_THREADS / 64 + 4) * 4 + 32;
                LSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_deep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_sort_deep, dim3(sg), dim3(SORT_THREADS), lds, st, a, r_cap);
            } else {
                hipLaunchKernelGGL(k_split_deep, dim3(sg), dim3(SPLIT_THREADS), 0, st, a);
            }
        }
        if (a.index_path) hipLaunchKernelGGL(k_cut, dim3((unsigned)(((uint64_t)c->n_slots * (NSLICE + 1) + 255) / 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_group_block, dim3((unsigned)(c->n_cus * 4)), dim3(BLOCK_THREADS), 0, st, a);
    }
    LSG_HIP(hipEventRecord(c->ev[1], st));
    if (n_ne > 0) {
        hipLaunchKernelGGL(k_walk_block<true>, dim3(grid_walk), dim3(WALK_THREADS), 0, st, a);
        hipLaunchKernelGGL(k_walk_block<false>, dim3((unsigned)c->n_cus), dim3(WALK_THREADS), 0, st, a);      // slots of more than WALK_PLANE_MAX entries: usually none
        LSG_HIP(hipEventRecord(c->ev[4], st));
        hipLaunchKernelGGL(k_pileup_huge, dim3(grid_block), dim3(BLOCK_THREADS), 0, st, a);
        if (c->n_multi > 0)
            hipLaunchKernelGGL(k_finalize_multi, dim3(c->n_multi < (unsigned)(c->n_cus * 8) ? c->n_multi : (unsigned)(c->n_cus * 8)), dim3(FIN_THREADS), 0, st, a);
    }
    LSG_HIP(hipEventRecord(c->ev[5], st));
    LSG_HIP(hipGetLastError());

    if (read_scalars(c, sc)) return -1;
    if (sc[SC_OVERFLOW]) { set_error("lsg_pileup_count: row buffer overflow (cap %llu)", (unsigned long long)c->row_cap); return -3; }
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->n_rows[i] = (int64_t)sc[SC_ROWS + i];
    c->n_columns = (int64_t)sc[SC_COLS];
    c->stats.n_reads_admitted = (int64_t)sc[SC_READS];
    c->stats.n_segs_admitted = (int64_t)sc[SC_SEGS];
    c->stats.n_events_admitted = (int64_t)sc[SC_EVENTS];
    c->stats.n_units = n_ne;
    c->stats.n_deep_units = (int64_t)c->n_slots - (int64_t)(sc[SC_NSMALL] & 0xffffffffull);
    c->stats.n_entries = n_ne > 0 ? total_entries : 0;
    // ms_bin: admission, scatter, plan, the small units and the deep units' sort and grouping, up to the walk's launch;
    // ms_deep: k_walk_block + k_pileup_huge + k_finalize_multi; ms_wave: k_pileup_wave (inside ms_bin)
    float ms = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->stats.ms_bin = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[1], c->ev[5])); c->stats.ms_deep = ms;
    c->stats.ms_walk = 0; c->stats.ms_wave = 0;
    if (n_ne > 0) {
        LSG_HIP(hipEventElapsedTime(&ms, c->ev[1], c->ev[4])); c->stats.ms_walk = ms;
        LSG_HIP(hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->stats.ms_wave = ms;
    }
    for (int i = 0; i < 4; ++i) { c->stats.rows_by_kernel[i] = (int64_t)sc[SC_ROWS_SRC + i]; c->stats.events_by_kernel[i] = (int64_t)sc[SC_EV_SRC + i]; }
    if (c->index_path) {      // no grouping kernel counted the walk's events: they are what the small units and the huge path did not read
        c->stats.events_by_kernel[1] = (int64_t)sc[SC_EVENTS] - (int64_t)sc[SC_EV_SRC + 0] - (int64_t)sc[SC_EV_SRC + 2];
        sc[SC_EV_DEEP] = (unsigned long long)c->stats.events_by_kernel[1] + sc[SC_EV_SRC + 2];
    }
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[5])); c->stats.ms_total = ms;
    c->stats.n_events_wave = (int64_t)sc[SC_EV_WAVE]; c->stats.n_events_deep = (int64_t)sc[SC_EV_DEEP];
    c->stats.n_rows_deep = (int64_t)sc[SC_ROWS_DEEP];
    c->stats.n_rows_wave = -c->stats.n_rows_deep;
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->stats.n_rows_wave += c->n_rows[i];
    c->last_params = *p;
    c->counted = true;
    c->called = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Canonical (genomic-order) export of one cell type's rows.
__global__ void k_unit_rowcount(const uint32_t* ne_units, const uint64_t* ne_mask, uint32_t n_ne, int n_ct, int ct, uint32_t* cnt) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > n_ne) return;
    uint32_t v = 0;
    if (w < n_ne && (int)(ne_units[w] % (uint32_t)n_ct) == ct) v = (uint32_t)__popcll(ne_mask[w]);
    cnt[w] = v;
}

__global__ void k_export_rows(CountArgs a, int ct, const uint32_t* rowoff, int64_t* keys, uint8_t* refs, uint32_t* counts) {
    const int lane = threadIdx.x & 63;
    uint32_t w = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= a.n_ne) return;
    const int2 geom = a.ne_geom[w];
    if ((int)((uint32_t)geom.y >> 24) != ct) return;
    const int tid = geom.y & 0xffffff;
    uint64_t em = a.ne_mask[w];
    if (!((em >> lane) & 1ull)) return;
    const uint32_t rb = a.ne_rowbase[w];
    uint64_t src = (uint64_t)(rb & ~ROW_NARROW) + __popcll(em & ((1ull << lane) - 1ull));
    uint64_t dst = (uint64_t)rowoff[w] + __popcll(em & ((1ull << lane) - 1ull));
    int64_t pos = (int64_t)geom.x + lane;
    keys[dst] = ((int64_t)tid << 32) | pos;
    refs[dst] = a.ref_ptr[tid][pos];
    if (rb & ROW_NARROW) {
        const uint16_t* h = reinterpret_cast<const uint16_t*>(a.rows[ct] + (src >> 6) * ROW_BLOCK_WORDS);
        for (int k = 0; k < ROW_PLANES; ++k) counts[dst * LSG_ROW_WORDS + k] = h[(k >> 2) * 256 + (src & 63) * 4 + (k & 3)];
    } else {
        for (int k = 0; k < ROW_PLANES; ++k) counts[dst * LSG_ROW_WORDS + k] = a.rows[ct][row_word(src, k)];
    }
    for (int sy = 0; sy < 8; ++sy)                             // BCr = BC - BCf is not stored
        counts[dst * LSG_ROW_WORDS + 34 + sy] = counts[dst * LSG_ROW_WORDS + 10 + sy] - counts[dst * LSG_ROW_WORDS + 26 + sy];
}

int run_fetch_counts(lsg_ctx* c, int ct, int64_t* keys, uint8_t* ref, uint32_t* counts, int64_t capacity) {
    if (!c->counted) { set_error("lsg_fetch_counts: call lsg_pileup_count first"); return -2; }
    if (ct < 0 || ct >= c->n_ct) { set_error("lsg_fetch_counts: bad cell type %d", ct); return -2; }
    int64_t n = c->n_rows[ct];
    if (capacity < n) { set_error("lsg_fetch_counts: capacity %lld < %lld rows", (long long)capacity, (long long)n); return -2; }
    if (n == 0) return 0;
    hipStream_t st = c->stream;
    CountArgs a{};
    fill_args(c, &c->last_params, a);
    uint32_t n_ne = c->n_ne;
    if (c->d_ne_rowoff.reserve((size_t)(n_ne + 2) * 8)) return -1;
    uint32_t* cnt = c->d_ne_rowoff.as<uint32_t>();
    uint32_t* off = cnt + (n_ne + 2);
    hipLaunchKernelGGL(k_unit_rowcount, dim3((n_ne + 256) / 256), dim3(256), 0, st, a.ne_units, a.ne_mask, n_ne, c->n_ct, ct, cnt);
    SCAN_U32(cnt, off, n_ne + 1);
    DevBuf &dk = c->ws[WS_EXPORT_K], &dr = c->ws[WS_EXPORT_R], &dc = c->ws[WS_EXPORT_C];
    if (dk.reserve((size_t)n * 8) || dr.reserve((size_t)n) || dc.reserve((size_t)n * LSG_ROW_WORDS * 4)) return -1;
    uint64_t threads = (uint64_t)n_ne * 64;
    hipLaunchKernelGGL(k_export_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, ct, off,
                       dk.as<int64_t>(), dr.as<uint8_t>(), dc.as<uint32_t>());
    LSG_HIP(hipMemcpyAsync(keys, dk.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(ref, dr.p, (size_t)n, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(counts, dc.p, (size_t)n * LSG_ROW_WORDS * 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    return 0;
}

// Upper bound of tile entries (sum over segments of tiles overlapped), computed once at load time.
__global__ void k_entries_upper(const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, unsigned long long* out) {
    unsigned long long v = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_segs; s += (int64_t)gridDim.x * blockDim.x) {
        int32_t st = seg_start[s], ln = seg_len[s];
        if (st >= 0 && ln > 0) v += (unsigned long long)(((uint32_t)(st + ln - 1) >> 6) - ((uint32_t)st >> 6) + 1);
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

int compute_entries_upper(lsg_ctx* c) {
    c->tile_caps_valid = false; c->index_valid = false; c->tm_valid = false; c->layout_build_ms = 0; c->seen_counts = 0;      // new reads: static capacities, the tile index and the tile-major store are rebuilt by the next count
    if (c->d_scalars.reserve(SC_COUNT * 8)) return -1;
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, c->stream));
    int64_t S = c->rd.n_segs;
    if (S > 0)
        hipLaunchKernelGGL(k_entries_upper, dim3((unsigned)((S + 255) / 256 < 2048 ? (S + 255) / 256 : 2048)), dim3(256), 0, c->stream, c->rd.seg_start,
                           c->rd.seg_len, S, c->d_scalars.as<unsigned long long>());
    unsigned long long v = 0;
    LSG_HIP(hipMemcpyAsync(&v, c->d_scalars.p, 8, hipMemcpyDeviceToHost, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    if (v >= 0x7FFFFFF0ull) { set_error("lsg_load_reads: %llu tile entries exceed the 31-bit entry index (two entry buffers); load the reads in windows", v); return -2; }
    c->entries_upper = (uint64_t)v;
    return 0;
}

} // namespace lsg
