// Per-cell-type pileup base counting on CDNA4 (gfx950) over the tile store (store.hip).
//
// Replaces, for every covered column at once:
//   split_bam's read routing            workflow/scripts/PreProcessing/SplitBamCellTypes.py:65-124
//   run_interval (pileup + counting)    workflow/scripts/SNVCalling/BaseCellCounter.py:182-320
//
// ONE form of the count.  The store holds every (segment x 64-position tile) entry of the load, a tile's entries adjacent and sorted
// by barcode, eight entries to a transposed 1 KB block; what a count decides per entry is admission (SAM flag, MAPQ, the pileup's
// max_depth drops: THIS count's parameters) and the cell type of its barcode (THIS table):
//   k_tm_resolve   one 32-bit meta word per entry, laid out so that the walk uses its fields as operands
//   k_tm_walk      workgroup = two waves = one job (a tile, or a run-aligned piece of a deep one): streams the job's blocks with 16-byte
//                  loads (lane = position, eight entries per load), adds into two pairs of LDS planes (both cell types of the pass in
//                  one sweep: a barcode's run belongs to one cell type), distinct-cell numbers = counts minus duplicates inside a run;
//                  then each wave finishes one cell type's unit (gates + rows) or writes its partial sums to the job's slab
//   k_tm_walk_wide the same for a job a single barcode's run stretches past what the packed planes hold (32-bit planes, plain C++)
//   k_finalize_multi  adds the slabs of multi-job tiles, gates, rows
// More than two cell types: one resolve + walk pass per pair of cell types.  No global atomics on the event path, integer arithmetic
// only (HBM- and issue-bound; no MFMA).
#include "lsg_ctx.h"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lsg {

#define LSG_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(LSG_AS3 const void*)p; }
constexpr int ARENA = 256;           // rows reserved per wave per allocation: this many or a multiple (lsg_ctx::arena)

// device scalars (uint64 each)
// The words kernels hammer while they run (work queue, row allocators) sit 128 bytes apart: atomics on one cache line serialise at
// ~90 per microsecond whichever of its words they name.
enum { SC_COLS = 8, SC_OVERFLOW = 9, SC_READS = 10, SC_SEGS = 11, SC_EVENTS = 12, SC_ROWS_DEEP = 15,
       SC_ROWS = 16, SC_ROWS_SRC = 28, SC_NENT = 38, SC_QWALK = 48,
       SC_ROWALLOC = 160, SC_ROWALLOC_STRIDE = 16, SC_COUNT = SC_ROWALLOC + SC_ROWALLOC_STRIDE * LSG_MAX_CELLTYPES };   // ROWS_SRC[4]: 1 k_tm_walk, 2 k_tm_walk_wide, 3 k_finalize_multi

struct CountArgs {
    // reads
    int64_t n_reads;
    const int32_t* read_tid; const uint16_t* read_flag; const uint8_t* read_mapq; const int32_t* read_cb;
    // genome / barcodes
    const int64_t* contig_len; const uint8_t* const* ref_ptr;
    const uint8_t* celltype_of;
    int32_t n_contigs, n_cb, n_ct;
    uint32_t tile_lo, tile_hi;          // counted tile range (lsg_set_region)
    // params
    int32_t min_bq, min_mq, min_dp, min_cc, ignore_orphans;
    uint32_t flag_exclude;
    const uint8_t* read_drop;             // the pileup's max_depth rule per read (layout.hip depth_cap_drops): 1 dropped in every window it overlaps, 2 in some; or null
    const unsigned long long* drop_pairs; int64_t n_drop_pairs;      // ... which: sorted (read << 32 | window of its contig)
    const uint32_t* tile_base; int32_t window;                       // first tile of every contig; the reference's pileup windows
    unsigned long long* adm;              // a bit per read: admitted under THIS count's read filters and not dropped (k_read_stats writes it, k_tm_resolve looks the
                                          // entries' reads up in it), or null when every stored read is admitted: the load filter already was this count's filter and nothing is dropped
    // units of the plan (copies the call stage and the exports read), rows
    uint32_t* ne_units; uint32_t* ne_nslot; uint32_t* ne_acc; int2* ne_geom;
    uint64_t* ne_mask; uint32_t* ne_rowbase;
    uint32_t* multi_list; uint32_t* macc;
    uint32_t n_ne, n_multi;
    uint32_t arena;                       // rows a wave reserves per allocation (multiple of 256)
    unsigned long long* scalars;
    uint32_t* rows[LSG_MAX_CELLTYPES];
    uint64_t row_cap;
};

// ------------------------------------------------------------------------------------------------
// Read admission = the union of the reference's filters on the count path:
//   pysam pileup flag_filter (UNMAP|SECONDARY|QCFAIL|DUP) and min_mapping_quality, ignore_orphans
//   (BaseCellCounter.py:191), is_supplementary (:249), CB tag present (:240-243), CB in barcodes.tsv
//   with a cell type (SplitBamCellTypes.py:83-90), MAPQ >= min_MQ (:110-113).
// The read-level half (SAM flag, MAPQ, orphans, the depth cap's drops) is decided HERE, once per read and count, and left as a bit per
// read; the entries of the store carry their read's index and k_tm_resolve looks the bit up (the barcode's cell type is the entry's own
// business).  The pass also counts the admitted READS (a statistic: lsg_count_stats.n_reads_admitted, the 24 B/read term of the
// algorithmic bytes).  Lanes of a wave take consecutive reads, the wave's first one a multiple of 64: one ballot = one word of the bitmap.
__global__ __launch_bounds__(256) void k_read_stats(CountArgs a) {
    unsigned long long n_ok = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t flag = a.read_flag[r];
        const int32_t cb = a.read_cb[r], tid = a.read_tid[r];
        bool pass = (flag & a.flag_exclude) == 0 && (int)a.read_mapq[r] >= a.min_mq;
        if (pass && a.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) pass = false;
        if (pass && a.read_drop && a.read_drop[r] == 1) pass = false;     // bam.pileup(..., max_depth): never entered the buffer of any window it overlaps
        if (a.adm) {
            const unsigned long long m = __ballot(pass);
            if ((threadIdx.x & 63) == 0) a.adm[r >> 6] = m;
        }
        bool ok = pass && cb >= 0 && cb < a.n_cb && tid >= 0 && tid < a.n_contigs;
        if (ok && a.celltype_of[cb] >= (uint32_t)a.n_ct) ok = false;
        n_ok += ok;
    }
    __shared__ unsigned long long s_ok;
    if (threadIdx.x == 0) s_ok = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) n_ok += __shfl_down(n_ok, o);
    if ((threadIdx.x & 63) == 0 && n_ok) atomicAdd(&s_ok, n_ok);
    __syncthreads();
    if (threadIdx.x == 0 && s_ok) atomicAdd(&a.scalars[SC_READS], s_ok);       // one same-address global atomic per workgroup
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t bq_threshold(const CountArgs& a) {
    const int q = a.min_bq < 0 ? 0 : (a.min_bq > 256 ? 256 : a.min_bq);
    return 0x800u + (uint32_t)q;
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
// Per-lane (= per reference position) sums of one unit.  Distinct-cell numbers come from duplicates: within a barcode run, an entry
// whose symbol was already seen at this position is a duplicate; CC[sym] = BC[sym] - dup[sym], NC = sum(BC) - ncdup
// (= len(set(...)), BaseCellCounter.py:283,292).
struct Acc {
    uint32_t bc[8], bq[8], bcf[8], dup[8], ncdup;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < 8; ++s) bc[s] = bq[s] = bcf[s] = dup[s] = 0;
        ncdup = 0;
    }
    __device__ __forceinline__ uint32_t BC(int s) const { return bc[s]; }
    __device__ __forceinline__ uint32_t BQ(int s) const { return bq[s]; }
    __device__ __forceinline__ uint32_t BCF(int s) const { return bcf[s]; }
    __device__ __forceinline__ uint32_t DUP(int s) const { return dup[s]; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};

// multi-job tiles: sum the unit's partial-sum slabs (8 waves, each a stride of the jobs), gates + emission by wave 0
constexpr int FIN_THREADS = 512;
__global__ __launch_bounds__(FIN_THREADS) void k_finalize_multi(CountArgs a) {
    __shared__ uint32_t sacc[NCTR][64];
    __shared__ WaveBook book;             // rows from a workgroup arena, exact counters flushed once (not five global atomics per unit)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (wv == 0) { book_init(book, lane); if (lane == 0) book.src = 3; }
    for (uint32_t k = blockIdx.x; k < a.n_multi; k += gridDim.x) {
        const uint32_t w = a.multi_list[k];
        const uint32_t nslot = a.ne_nslot[w];
        { const uint32_t tile = a.ne_units[w] / (uint32_t)a.n_ct; if (tile < a.tile_lo || tile >= a.tile_hi) continue; }   // (the plan's lists are static: all tiles)
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

static int read_scalars(lsg_ctx* c, unsigned long long* sc) {
    // pinned landing zone: a pageable destination costs a staging copy and tens of microseconds per read
    LSG_HIP(hipMemcpyAsync(c->h_pin, c->d_scalars.p, SC_COUNT * 8, hipMemcpyDeviceToHost, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    memcpy(sc, c->h_pin, SC_COUNT * 8);
    return 0;
}

static void fill_args(lsg_ctx* c, const lsg_count_params* p, CountArgs& a) {
    a.n_reads = c->rd.n_reads;
    a.read_tid = c->rd.read_tid; a.read_flag = c->rd.read_flag; a.read_mapq = c->rd.read_mapq; a.read_cb = c->rd.read_cb;
    a.contig_len = c->d_contig_len.as<int64_t>();
    a.ref_ptr = c->d_ref_ptrs.as<const uint8_t*>(); a.celltype_of = c->d_celltype_of.as<uint8_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.n_ct = c->n_ct;
    a.tile_lo = c->tile_lo; a.tile_hi = c->tile_hi;
    a.min_bq = p->min_bq; a.min_mq = p->min_mq; a.min_dp = p->min_dp; a.min_cc = p->min_cc;
    a.ignore_orphans = p->ignore_orphans; a.flag_exclude = p->flag_exclude;
    a.read_drop = c->has_drops ? c->d_read_drop.as<uint8_t>() : nullptr;
    a.drop_pairs = c->has_drops && c->n_drop_pairs ? c->d_drop_pairs.as<unsigned long long>() : nullptr; a.n_drop_pairs = c->n_drop_pairs;
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.window = c->st_window;
    // every stored read passed the load filter: a count under exactly that filter, with nothing dropped, admits them all
    const bool all_in = !c->has_drops && p->min_mq <= c->st_min_mq && (p->flag_exclude & ~c->st_flag_exclude) == 0 && (!p->ignore_orphans || c->st_ignore_orphans);
    a.adm = all_in ? nullptr : c->d_read_adm.as<unsigned long long>();
    a.ne_units = c->d_ne_units.as<uint32_t>(); a.ne_nslot = c->ws[WS_NE_NSLOT].as<uint32_t>();
    a.ne_acc = c->ws[WS_NE_ACC].as<uint32_t>(); a.ne_geom = c->ws[WS_NE_GEOM].as<int2>();
    a.ne_mask = c->d_ne_mask.as<uint64_t>(); a.ne_rowbase = c->d_ne_rowbase.as<uint32_t>();
    a.multi_list = c->ws[WS_MULTI_LIST].as<uint32_t>(); a.macc = c->ws[WS_MACC].as<uint32_t>();
    a.n_ne = c->n_ne; a.n_multi = c->n_multi;
    a.scalars = c->d_scalars.as<unsigned long long>();
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) a.rows[i] = c->d_rows[i].as<uint32_t>();
    a.row_cap = c->row_cap; a.arena = c->arena;
}

// launch-shape knobs for tuning runs (environment overrides; the defaults are what ships)
static int tune_int(const char* name, int dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const int x = atoi(v);
    return x > 0 && x <= 64 ? x : dflt;
}

#define SCAN_U32(in, out, n)                                                                              \
    do {                                                                                                  \
        size_t tb_ = 0;                                                                                   \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_, (in), (out), (int)(n), st));              \
        if (c->d_cub_tmp.reserve(tb_ + 256)) return -1;                                                   \
        tb_ = c->d_cub_tmp.cap;                                                                           \
        LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb_, (in), (out), (int)(n), st));       \
    } while (0)

// ================================================================================================
// A count over the tile store.
//   meta[p] (per count and pass, 32 bits laid out so that the walk uses them as operands): second cell type of the pass << 4 and << 12 |
//           forward << 20 | not counted or not there << 29 | run of one entry << 30 | run start << 31
#ifndef LSG_TM_ASM
#define LSG_TM_ASM true
#endif
constexpr int IX_STAT_SLOTS = 256;
constexpr uint32_t TMM_CT4 = 1u << 4, TMM_CT12 = 1u << 12, TMM_FWD = 1u << 20, TMM_SKIP = 1u << 29, TMM_SINGLE = 1u << 30, TMM_RS = 1u << 31;
struct TmArgs {
    const uint4* store; const uint32_t* s0; const uint8_t* b; const uint32_t* rd; uint32_t* meta; const uint32_t* blk_tile; const TmJob* jobs;
    const uint32_t* chunk_start;          // static: first job of every chunk of about TM_CHUNK_WORK work
    const uint16_t* ext;                  // static, per block: first position any of its entries has an event at | one past the last << 8
    uint64_t np; uint32_t nblk, njobs, nchunks;
    int32_t ct_base;                      // the pass counts cell types ct_base and ct_base + 1
};

// the pileup's depth cap dropped read r in SOME of the windows it overlaps (read_drop[r] == 2): in the one this entry lies in?  An entry never
// crosses a window edge (store.hip): its window is the one its tile starts in, or - bit TM_WHI of s0 - the one that starts inside the tile.
__device__ __noinline__ bool tm_dropped_here(const CountArgs& a, uint32_t r, uint32_t tile, uint32_t s0word) {
    int lo = 0, hi = a.n_contigs;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.tile_base[mid] <= tile) lo = mid; else hi = mid; }
    const int64_t tstart = (int64_t)(tile - a.tile_base[lo]) << 6;
    const uint64_t w = (uint64_t)((tstart >= 1 ? (tstart - 1) / a.window : 0) + ((s0word & TM_WHI) ? 1 : 0));
    const unsigned long long key = ((unsigned long long)r << 32) | w;
    int64_t b = 0, e = a.n_drop_pairs;
    while (b < e) { const int64_t m = (b + e) >> 1; if (a.drop_pairs[m] < key) b = m + 1; else e = m; }
    return b < a.n_drop_pairs && a.drop_pairs[b] == key;
}

// per count and pass: the word the walk reads per entry.  Admission under THIS count's parameters (the entry's read: its bit of
// k_read_stats' bitmap - SAM flag, MAPQ, the depth cap's drops), cell type under THIS barcode table, region.
// DROPS: some reads are dropped by the depth cap in SOME of their windows (the look-up is a call: kept out of the common instantiation)
template <bool DROPS>
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
        uint32_t rv[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        if (in_region && a.adm) {
            const uint4* rp = reinterpret_cast<const uint4*>(tm.rd + (uint64_t)blk * 8);
            const uint4 r_lo = rp[0], r_hi = rp[1];
            rv[0] = r_lo.x; rv[1] = r_lo.y; rv[2] = r_lo.z; rv[3] = r_lo.w; rv[4] = r_hi.x; rv[5] = r_hi.y; rv[6] = r_hi.z; rv[7] = r_hi.w;
        }
        uint32_t mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t s = sv[u], b8 = ((u < 4 ? bb.x : bb.y) >> (8 * (u & 3))) & 0xffu, cb = s & CB_MASK;
            uint32_t cls = 3;
            if (in_region) {
                cls = 2;
                bool ok = cb < (uint32_t)a.n_cb;             // (pad entries carry CB_MASK)
                if (ok && a.adm) ok = (reinterpret_cast<const uint32_t*>(a.adm)[rv[u] >> 5] >> (rv[u] & 31u)) & 1u;
                if (DROPS) { if (ok && a.read_drop[rv[u]] == 2) ok = !tm_dropped_here(a, rv[u], tile, s); }
                if (ok) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
            }
            if (cls < 2) { ev += (b8 & 63u) + 1u; sg += (b8 >> 6) & 1u; ++ne; }
            uint32_t m = cls < 2 ? (cls ? (TMM_CT4 | TMM_CT12) : 0u) | ((s & TM_FWD) ? TMM_FWD : 0u) | ((b8 & 128u) ? TMM_SINGLE : 0u) : TMM_SKIP;
            if (cls != 3 && (s & TM_RUNSTART)) m |= TMM_RS;              // (an entry that is not there starts nothing)
            mv[u] = m;
        }
        uint4* mp = reinterpret_cast<uint4*>(tm.meta + (uint64_t)blk * 8);
        mp[0] = make_uint4(mv[0], mv[1], mv[2], mv[3]); mp[1] = make_uint4(mv[4], mv[5], mv[6], mv[7]);
    }
    for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
    if ((threadIdx.x & 63) == 0 && ne) { atomicAdd(&s_stat[0], ev); atomicAdd(&s_stat[1], sg); atomicAdd(&s_stat[2], ne); }
    __syncthreads();
    if (threadIdx.x == 0 && s_stat[2] && stat_slots) {
        unsigned long long* slot = stat_slots + (size_t)(blockIdx.x % IX_STAT_SLOTS) * 8;      // 64 bytes apart: one word takes ~90 atomics per microsecond
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
// AL (k_tm_count_direct over tile-phased events): the lane's event register was loaded from the entry's whole 128-byte LINE, so the lanes
// outside the entry's positions [first, end) hold somebody else's events: the meta word carries first in its bits 0..5 and 64 - end in its
// bits 13..18 (its cell-type bit moves from bit 4 to bit 10, TMM_CT10) and EXEC is cut to the entry's lanes before the lanes that count are selected; thr is
// the base quality itself, compared with the event's low byte in one operation (an event that is not there is 0: include/longsom_hip.h).
constexpr uint32_t TMM_CT10 = 1u << 10;
template <bool HI, bool AL = false>
__device__ __forceinline__ void tm_add(TmState& s, uint32_t m, uint32_t evw, uint32_t thr, uint32_t pkl0, uint32_t one) {
    if (!TM_ASM) {                                   // the same in plain C++ (what the asm block is checked against when it is touched)
        const uint32_t ev = HI ? evw >> 16 : evw & 0xffffu;
        if (m & TMM_CLOSE) { s.nc += s.mask & 0x10001u; s.mask = 0; }
        if (m & TMM_SKIP) return;
        const uint32_t al_lane = threadIdx.x & 63u, al_end = 64u - ((m >> 13) & 63u);
        const bool counted = AL ? (ev & 0xffu) >= thr && al_lane >= (m & 63u) && al_lane < al_end : (ev & 0x8ffu) >= thr;
        const uint32_t addr = (pkl0 | (ev & 0x700u)) + (m & TMM_CT12);
        const uint32_t lo = (ev & 0xffu) | (m & TMM_FWD);
        const uint32_t sym8 = (ev >> 8) & 15u, ctone = AL ? 1u << ((m >> 6) & 16u) : 1u << (m & TMM_CT4);
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
    uint32_t t0, t1, addr, lo, sa, sb; unsigned long long xm;
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
    // the same for an entry of tile-phased events (AL, above): no mask of the event (the compare reads its low byte), EXEC cut to the
    // entry's lanes [first, end) by two shifts that take their amounts from the meta word (first: bits 0..5; 64 - end: bits 13..18),
    // the cell type's bit 10 shifted down to make 0 or 16
#define LSG_TM_ADD_AL(WSEL, BSEL, SYMPOS)                                                                                            \
    asm volatile(                                                                                                                    \
        "s_bitcmp1_b32 %[m], 27\n\t"                                                                                                  \
        "s_cbranch_scc0 1f\n\t"                                                                                                       \
        "v_and_b32 %[t1], 0x10001, %[mask]\n\t"                                                                                       \
        "v_add_u32 %[nc], %[nc], %[t1]\n\t"                                                                                           \
        "v_mov_b32 %[mask], 0\n"                                                                                                      \
        "1:\n\t"                                                                                                                      \
        "s_bitcmp1_b32 %[m], 29\n\t"                                                                                                  \
        "s_cbranch_scc1 5f\n\t"                                                                                                       \
        "s_and_b32 %[sa], %[m], 0x1000\n\t"                                                                                           \
        "v_and_b32_sdwa %[addr], %[c700], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" WSEL "\n\t"             \
        "s_lshr_b32 %[sb], %[m], 13\n\t"                                                                                              \
        "v_or3_b32 %[addr], %[addr], %[pkl], %[sa]\n\t"                                                                               \
        "s_lshl_b64 %[xm], -1, %[m]\n\t"                          /* the lanes from the entry's first position upwards ... */      \
        "s_lshr_b64 exec, -1, %[sb]\n\t"                          /* ... and below its end */                                      \
        "s_and_b32 %[sb], %[m], 0x100000\n\t"                                                                                         \
        "s_and_b64 exec, exec, %[xm]\n\t"                                                                                             \
        "v_or_b32_sdwa %[lo], %[sb], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BSEL "\n\t"                  \
        "s_lshr_b32 %[sa], %[m], 6\n\t"                                                                                               \
        "v_cmpx_le_u32_sdwa vcc, %[thr], %[ev] src0_sel:DWORD src1_sel:" BSEL "\n\t"   /* EXEC = the lanes that count this event */ \
        "s_lshl_b32 %[sa], 1, %[sa]\n\t"                                                                                              \
        "s_bitcmp1_b32 %[m], 30\n\t"                                                                                                  \
        "s_cbranch_scc0 2f\n\t"                                                                                                       \
        "ds_add_u32 %[addr], %[lo]\n\t"                                                                                               \
        "ds_add_u32 %[addr], %[one] offset:2048\n\t"                                                                                  \
        "v_add_u32 %[nc], %[sa], %[nc]\n\t"                                                                                           \
        "s_branch 4f\n"                                                                                                               \
        "2:\n\t"                                                                                                                      \
        "v_bfe_u32 %[t1], %[ev], " SYMPOS ", 4\n\t"                                                                                   \
        "s_bitcmp1_b32 %[m], 28\n\t"                                                                                                  \
        "s_cbranch_scc0 3f\n\t"                                                                                                       \
        "ds_add_u32 %[addr], %[lo]\n\t"                                                                                               \
        "ds_add_u32 %[addr], %[one] offset:2048\n\t"                                                                                  \
        "v_lshl_or_b32 %[mask], %[one], %[t1], %[sa]\n\t"                                                                             \
        "s_branch 4f\n"                                                                                                               \
        "3:\n\t"                                                                                                                      \
        "v_bfe_u32 %[t0], %[mask], %[t1], 1\n\t"                                                                                      \
        "v_lshl_or_b32 %[t0], %[t0], 16, %[one]\n\t"                                                                                  \
        "ds_add_u32 %[addr], %[lo]\n\t"                                                                                               \
        "ds_add_u32 %[addr], %[t0] offset:2048\n\t"                                                                                   \
        "v_lshl_or_b32 %[t1], %[one], %[t1], %[sa]\n\t"                                                                               \
        "v_or_b32 %[mask], %[mask], %[t1]\n"                                                                                          \
        "4:\n\t"                                                                                                                      \
        "s_mov_b64 exec, -1\n"                                                                                                        \
        "5:"                                                                                                                          \
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [addr] "=&v"(addr), [lo] "=&v"(lo), [sa] "=&s"(sa), [sb] "=&s"(sb), [xm] "=&s"(xm),        \
          [mask] "+v"(s.mask), [nc] "+v"(s.nc)                                                                                       \
        : [ev] "v"(evw), [m] "s"(m), [thr] "s"(thr), [c700] "s"(0x700u), [one] "v"(one), [pkl] "v"(pkl0)                              \
        : "scc", "vcc", "memory")
    if (AL) { if (HI) LSG_TM_ADD_AL("WORD_1", "BYTE_2", "24"); else LSG_TM_ADD_AL("WORD_0", "BYTE_0", "8"); }
    else if (HI) LSG_TM_ADD("WORD_1", "BYTE_2", "24"); else LSG_TM_ADD("WORD_0", "BYTE_0", "8");
#undef LSG_TM_ADD
#undef LSG_TM_ADD_AL
}
struct TmCounters {
    const uint32_t* pl; int lane; uint32_t ncdup;
    __device__ __forceinline__ uint32_t BC(int k) const { return pl[512 + k * 64 + lane] & 0xffffu; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return pl[512 + k * 64 + lane] >> 16; }
    __device__ __forceinline__ uint32_t BQ(int k) const { return pl[k * 64 + lane] & 0xfffffu; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return pl[k * 64 + lane] >> 20; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};

constexpr int TMW_WAVES = 2;       // two waves share a job (and its planes); TM_GROUP blocks per load group: 4 KB in flight per wave and group
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
    __shared__ __attribute__((aligned(8192))) uint32_t planes[2][2][8 * 64];      // [cell type of the pass][plane][symbol row x lane]: the cell type is bit 12 of an address
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
        if (threadIdx.x == 0) s_ck = first ? blockIdx.x : (uint32_t)atomicAdd(&a.scalars[SC_QWALK], 1ull) + gridDim.x;
        __syncthreads();
        const uint32_t ck = rl(s_ck, 0);
        if (ck >= tm.nchunks) break;
        const uint32_t jx_end = rl(tm.chunk_start[ck + 1], 0);
        for (uint32_t jx = rl(tm.chunk_start[ck], 0); jx < jx_end; ++jx) {
            uint32_t jw = 0;
            if (lane < 8) jw = reinterpret_cast<const uint32_t*>(tm.jobs + jx)[lane];
            const uint32_t e0 = rl(jw, 0), e1 = rl(jw, 1), w0 = rl(jw, 2), slab = rl(jw, 3), nj = rl(jw, 4), tcnt = rl(jw, 5), tile = rl(jw, 6), emid = rl(jw, 7);
            if (tile < a.tile_lo || tile >= a.tile_hi || (nj & TMJ_WIDE)) continue;          // (a job too long for these planes: k_tm_walk_wide)
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
            // the tile's units of this pass: wave = cell type
            const int ct = tm.ct_base + wv;
            if (ct < a.n_ct) {
                const uint32_t* pc = pl + wv * 1024;
                uint32_t dp = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp += pc[512 + k * 64 + lane] & 0xffffu;
                const TmCounters tot{pc, lane, dp - nc_sh[wv][lane]};
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

// ================================================================================================
// The load's gather and the FIRST count in one pass (lsg_set_count_at_load).  A rule of the reference counts a BAM once
// (BaseCellCounter.py:182-320); when the count's parameters and the barcode table are known while the reads are loaded, the wave that
// transposes a block of the store (store.hip k_tm_gather: eight entries' events from the caller's compact array -> [position][entry])
// holds exactly the registers the walk would fetch again from HBM a moment later, so it adds them into the job's planes right there:
// no k_tm_resolve, no second pass over the 24 GB of blocks.  Same jobs, planes, run logic (tm_add), units and rows as k_tm_walk; what
// differs is where a group's rows come from.  Per group of TM_GROUP blocks a wave
//   K  loads the sorted (key, value) of its 32 entries in place (lanes 0..31; lanes 32 / 33: the keys on either side of the group)
//   W  makes the store's per-entry words s0, b, rd (run flags from the neighbouring keys' barcodes) and THIS count's meta word
//      (admission bit of the entry's read, cell type of its barcode: what k_tm_resolve would write)
//   E  fetches the events: lane = (entry u of a block, 16-byte chunk c), the chunk ALIGNED TO THE TILE (events at positions 8c .. 8c+7,
//      from source offset - first position + 8c), so that an entry's row of the LDS tile is indexed by position
//   T  crosses the LDS tile: lane = position reads one event per entry, kept where the entry's position mask (an SGPR pair) says so;
//      the rows leave as the block's kilobyte (non-temporal) AND stay in registers for the walk's consume step
// software-pipelined: the keys of group g + 2 and the events of group g + 1 are in flight while group g is counted.
// A block whose entries belong to two ranges (a job's cut is not a multiple of eight) is written by the range its first entry lies in;
// the other range gathers it too, for its own entries' events only.  Jobs outside the counted region or too long for the packed planes
// (TMJ_WIDE) are gathered without counting (the wide walk takes the latter from the store afterwards).
struct TgArgs {
    const uint16_t* events; int64_t n_events;
    const uint64_t* key; const uint32_t* rdv; int32_t cb_bits;      // the sort's output (store.hip sort_key): read in place
    uint64_t src_mask;                                                // the bits of the key's source field (a load that sorts keys alone - rdv null - keeps the entry's two flags above them)
    int32_t src_shift;                                                // 6: tile-phased events (LSG_LAYOUT_PHASED) - the source field is the entry's 128-byte line, the event index = line << 6 | first position; 0: the event index itself
    const uint32_t* tile_off; const uint32_t* blk_off;
    uint32_t* s0; uint8_t* b8; uint32_t* rd; uint4* store; uint16_t* ext;
    const uint32_t* nchunks;                                          // the plan's number of chunks, still on the device
    unsigned long long* queues;                                       // k_tm_count_direct: the XCDs' job queues, 128 bytes apart (zeroed before the launch)
    unsigned long long* stat_slots;
};
typedef uint32_t tg_u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) TgU4A2 { tg_u32x4 v; };
constexpr uint32_t TG_RV_WHI = 1u << 29, TG_RV_SEGFIRST = 1u << 30, TG_RV_FWD = 1u << 31, TG_RV_READ = TG_RV_WHI - 1u;      // (store.hip RV_*)
struct TgKeys { uint64_t k; uint32_t v; };
struct TgStat { uint32_t ev, sg, ne; };
struct TgPre { uint32_t bits, ctv, admw; };      // an entry's meta word in the making: flags | bit of its read | events, and the two table words looked up for it

// ---- K  the sorted (key, value) of the 32 entries from padded entry p0 on (lanes 0..31), and the keys on either side (lanes 32, 33).
//         Every lane loads, lanes past 33 and entries that are not there from a clamped place: loads outside branches let the compiler
//         count them, so that waiting for one group's data leaves the next group's in flight.
__device__ __forceinline__ TgKeys tg_load_keys(const TgArgs& tg, uint32_t p0, uint32_t base, uint32_t off, uint32_t n, int lane) {
    TgKeys r;
    const int64_t i = (int64_t)p0 - (int64_t)base + (lane < 32 ? lane : (lane == 32 ? -1 : 32));
    const uint32_t ic = i < 0 ? 0u : (i >= (int64_t)n ? n - 1u : (uint32_t)i);
    r.k = __builtin_nontemporal_load(tg.key + off + ic);
    r.v = __builtin_nontemporal_load(tg.rdv + off + ic);
    return r;
}

// K0: the keys of the range's first group (the caller has them on their way: loaded while the job before was finishing)
__device__ __forceinline__ void tg_range(const CountArgs& a, const TmArgs& tm, const TgArgs& tg, TmState& st, TgStat& stat, uint32_t s0r, uint32_t s1r, uint32_t base, uint32_t off,
                                         uint32_t n, bool in_region, bool counting, uint32_t thr, uint32_t pkl0, uint32_t one, int lane, uint16_t (*xt)[8][64], const tg_u32x4 (*tmask)[9],
                                         TgKeys K) {
    const uint32_t b0 = s0r >> 3, nblk = ((s1r + 7) >> 3) - b0;
    const int ng = (int)((nblk + TM_GROUP - 1) / TM_GROUP);
    const bool tail = s1r == base + n;                                    // the range that ends its tile also writes the tile's pad entries
    const uint32_t cbm = (1u << tg.cb_bits) - 1u;
    const int eu = lane >> 3, ec = lane & 7;
    auto nt_put = [](auto* q, auto v) { __builtin_nontemporal_store(v, q); };
    auto load_keys = [&](int g) -> TgKeys { return tg_load_keys(tg, (b0 + (uint32_t)g * TM_GROUP) * 8, base, off, n, lane); };
    // ---- W + E: the entries' words, what the meta word needs (its two table look-ups are ISSUED here and read an iteration later:
    //      finish_meta), the blocks' extents; the event loads.  A chunk is fetched whole from the caller's array, so what it holds beside the
    //      entry's own events (the read's neighbouring segment) is cleared before it crosses the LDS tile: lh keeps, per block, the
    //      half-words [lo, hi) of the lane's chunk that are the entry's
    auto words_events = [&](int g, const TgKeys& K, tg_u32x4 (&chunk)[TM_GROUP], TgPre& pre, uint32_t& lh, uint32_t& xe) {
        const uint32_t p = (b0 + (uint32_t)g * TM_GROUP) * 8 + (uint32_t)lane, i = p - base;
        const bool valid = lane < 32 && i < n;
        const uint32_t cb = (uint32_t)K.k & cbm;
        const uint32_t cb_prev = (uint32_t)__shfl((int)cb, lane == 0 ? 32 : lane - 1), cb_next = (uint32_t)__shfl((int)cb, lane == 31 ? 33 : lane + 1);
        const uint32_t geom = (uint32_t)(K.k >> tg.cb_bits), first = valid ? geom & 63u : 0u, nev = valid ? ((geom >> 6) & 63u) + 1u : 0u;
        const uint64_t src = ((K.k >> (tg.cb_bits + 12)) << tg.src_shift) | (tg.src_shift ? geom & 63u : 0u);
        const bool rs = i == 0 || cb_prev != cb;
        const bool single = rs && (i + 1 == n || cb_next != cb);
        const bool own = valid && p >= s0r && p < s1r;
        if (own) {
            nt_put(tg.s0 + p, cb | ((K.v & TG_RV_FWD) ? TM_FWD : 0u) | ((K.v & TG_RV_WHI) ? TM_WHI : 0u) | (rs ? TM_RUNSTART : 0u));
            nt_put(tg.b8 + p, (uint8_t)((nev - 1u) | ((K.v & TG_RV_SEGFIRST) ? 64u : 0u) | (single ? 128u : 0u)));
            nt_put(tg.rd + p, K.v & TG_RV_READ);
        } else if (lane < 32 && !valid && tail && p < (b0 + nblk) * 8) {
            nt_put(tg.s0 + p, (uint32_t)TM_PAD_S0); nt_put(tg.b8 + p, (uint8_t)0); nt_put(tg.rd + p, 0u);
        }
        const uint32_t r = K.v & TG_RV_READ;
        // (the aligned word that holds the barcode's cell type: its byte is picked when the word is used, an iteration later - nothing here waits for it)
        pre.ctv = reinterpret_cast<const uint32_t*>(a.celltype_of)[(cb < (uint32_t)a.n_cb ? cb : 0u) >> 2];
        pre.admw = a.adm ? reinterpret_cast<const uint32_t*>(a.adm)[(r < (uint32_t)a.n_reads ? r : 0u) >> 5] : 0xffffffffu;
        pre.bits = (own && in_region && counting ? 1u : 0u) | (cb < (uint32_t)a.n_cb ? 2u : 0u) | ((K.v & TG_RV_FWD) ? 4u : 0u) | (single ? 8u : 0u) | (rs ? 16u : 0u) |
                   ((r & 31u) << 8) | (nev << 16) | ((K.v & TG_RV_SEGFIRST) ? (1u << 24) : 0u) | (own && in_region ? (1u << 25) : 0u) | ((cb & 3u) << 26);
        // a block's extent: the positions its entries' ranges cover (lanes 8 q hold block q's)
        uint32_t xlo = valid ? first : 64u, xhi = first + nev;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const uint32_t l2 = (uint32_t)__shfl_xor((int)xlo, o), h2 = (uint32_t)__shfl_xor((int)xhi, o);
            xlo = l2 < xlo ? l2 : xlo; xhi = h2 > xhi ? h2 : xhi;
        }
        xe = xlo | (xhi << 8);
        // where position 0 of the tile would lie in the caller's array (the entry's events start `first` further on)
        const int64_t abase = (int64_t)src - (int64_t)first;
        const uint32_t e_lo = (uint32_t)abase, e_inf = ((uint32_t)((uint64_t)abase >> 32) & 0xffu) | (first << 8) | ((first + nev) << 16);
        // the first / last events of the array: a group that holds one of them loads its chunks element by element, never outside the array
        const bool edge = __ballot(valid && (src < 64ull || (int64_t)src + 128 > tg.n_events)) != 0ull;
        lh = 0;
#pragma unroll
        for (int q = 0; q < TM_GROUP; ++q) {
            const uint32_t lo32 = (uint32_t)__shfl((int)e_lo, q * 8 + eu), inf = (uint32_t)__shfl((int)e_inf, q * 8 + eu);
            const int c8 = ec * 8, d0 = (int)((inf >> 8) & 0xffu) - c8, d1 = (int)(inf >> 16) - c8;
            const uint32_t lok = (uint32_t)(d0 < 0 ? 0 : (d0 > 8 ? 8 : d0)), hik = (uint32_t)(d1 < 0 ? 0 : (d1 > 8 ? 8 : d1));
            const bool need = lok < hik;                                      // the tile-aligned chunk holds events of the entry
            lh |= (lok | (hik << 4)) << (8 * q);
            const int64_t A = (int64_t)(int8_t)(inf & 0xffu) * (1ll << 32) + (int64_t)lo32 + c8;      // (abase may be a little below zero: its bits 32..39 sign-extended)
            if (__builtin_expect(!edge, 1)) {
                // chunks nobody needs read the array's first line (one cached line, no branch: the compiler can count the loads in flight)
                chunk[q] = reinterpret_cast<const TgU4A2*>(tg.events + (need ? A : 0))->v;
            } else {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (need)
                    for (int k = 0; k < 8; ++k) if (A + k >= 0 && A + k < tg.n_events) w[k >> 1] |= (uint32_t)tg.events[A + k] << (16 * (k & 1));
                chunk[q] = tg_u32x4{w[0], w[1], w[2], w[3]};
            }
        }
    };
    // what k_tm_resolve writes for the entry under this count (an entry of another range, or outside the counted region, is not there)
    auto finish_meta = [&](const TgPre& pre) -> uint32_t {
        uint32_t M = TMM_SKIP;
        if (pre.bits & (1u << 25)) {
            uint32_t cls = 2;
            bool ok = (pre.bits & 2u) != 0 && ((pre.admw >> ((pre.bits >> 8) & 31u)) & 1u);
            if (ok) { const uint32_t ct = (pre.ctv >> (((pre.bits >> 26) & 3u) * 8u)) & 0xffu; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
            if (cls < 2) { stat.ev += (pre.bits >> 16) & 0xffu; stat.sg += (pre.bits >> 24) & 1u; ++stat.ne; }
            if (pre.bits & 1u) {
                M = cls < 2 ? (cls ? (TMM_CT4 | TMM_CT12) : 0u) | ((pre.bits & 4u) ? TMM_FWD : 0u) | ((pre.bits & 8u) ? TMM_SINGLE : 0u) : TMM_SKIP;
                if (pre.bits & 16u) M |= TMM_RS;
            }
        }
        return M;
    };
    // ---- T: the chunks, cleared outside their entries, into the LDS tile ([entry][position]); then block by block: lane = position reads
    //      one event per entry (no index arithmetic, nothing to mask), the row goes to the store and, still in registers, to the counters
    auto tile_write = [&](tg_u32x4 (&chunk)[TM_GROUP], uint32_t lh) {
#pragma unroll
        for (int q = 0; q < TM_GROUP; ++q) {
            const tg_u32x4 keep = tmask[0][(lh >> (8 * q + 4)) & 15u] & tmask[1][(lh >> (8 * q)) & 15u];      // half-words below hi, and not below lo
            *reinterpret_cast<tg_u32x4*>(&xt[q][eu][ec * 8]) = chunk[q] & keep;
        }
        lds_fence();
    };
    auto block_row = [&](int g, int q, uint32_t xe) -> tg_u32x4 {
        const uint32_t blk = (uint32_t)g * TM_GROUP + q;                   // relative to b0 (the caller has checked blk < nblk)
        uint32_t e[8];
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) e[uu] = xt[q][uu][lane];
        const tg_u32x4 row = {e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        if ((b0 + blk) * 8 >= s0r) {                     // this range writes the block: the rows of its extent
            const uint32_t x = rl(xe, q * 8), first = x & 0xffu, last = x >> 8;
            if ((uint32_t)lane - first < last - first) __builtin_nontemporal_store(row, reinterpret_cast<tg_u32x4*>(tg.store) + (uint64_t)(b0 + blk) * 64 + lane);
            if (lane == 0) tg.ext[b0 + blk] = (uint16_t)x;
        }
        return row;
    };
    static_assert(8 * TM_GROUP == 32, "one 32-bit mask per group");
    uint32_t open_in = 0;
    auto verdicts = [&](uint32_t M) -> uint32_t {                           // (as in tm_walk_range)
        const bool there = !(M & TMM_SKIP), multi = there && !(M & TMM_SINGLE), rs = (M & TMM_RS) != 0;
        const uint32_t A = (uint32_t)__ballot(lane < 32 && multi), R = (uint32_t)__ballot(lane < 32 && rs);
        const uint32_t below = lane < 32 ? (1u << lane) - 1u : 0xffffffffu;
        const uint32_t rb = R & below;
        const uint32_t seg = rb ? below & ~((1u << (31 - __clz(rb))) - 1u) : below;
        const bool ob = (A & seg) != 0u || (!rb && open_in);
        if (rs && ob) M |= TMM_CLOSE;
        if (multi && (rs || !ob)) M |= TMM_FIRST;
        const uint32_t segl = R ? ~((1u << (31 - __clz(R))) - 1u) : 0xffffffffu;
        open_in = ((A & segl) != 0u || (!R && open_in)) ? 1u : 0u;
        return M;
    };
    auto consume_block = [&](int q, const tg_u32x4& row, uint32_t M) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t m = rl(M, q * 8 + u);
            if (u & 1) tm_add<true>(st, m, row[u >> 1], thr, pkl0, one);
            else tm_add<false>(st, m, row[u >> 1], thr, pkl0, one);
        }
    };
    // One set of registers per stage and this order: a stage's loads are issued after the previous use of their registers and first read an
    // iteration later (a copy of a freshly loaded register would make the wave wait for it - and for every load issued before it).  A
    // group's rows leave the LDS tile one block at a time: four registers of events alive beside the sixteen in flight.
    tg_u32x4 chunk[TM_GROUP];
    uint32_t lh = 0, xe = 0;
    TgPre pre;
    words_events(0, K, chunk, pre, lh, xe);
    K = load_keys(ng > 1 ? 1 : 0);
    for (int g = 0; g < ng; ++g) {
        tile_write(chunk, lh);                                // waits for the events of group g (issued an iteration ago)
        const uint32_t Mc = verdicts(finish_meta(pre));    // ... whose look-ups were issued with them
        const uint32_t xe_g = xe;
        if (g + 1 < ng) {
            words_events(g + 1, K, chunk, pre, lh, xe);
            K = load_keys(g + 2 < ng ? g + 2 : ng - 1);
        }
#pragma unroll
        for (int q = 0; q < TM_GROUP; ++q) {
            if ((uint32_t)g * TM_GROUP + q >= nblk) break;
            const tg_u32x4 row = block_row(g, q, xe_g);
            if (counting) consume_block(q, row, Mc);
        }
        lds_fence();                                          // the tile is read before the next group's chunks are written into it
    }
    if (counting) { st.nc += st.mask & 0x10001u; st.mask = 0; }
}

#ifndef LSG_TG_WAVES_PER_EU
#define LSG_TG_WAVES_PER_EU 5
#endif
__global__ __launch_bounds__(TMW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(LSG_TG_WAVES_PER_EU))) void k_tm_gather_count(CountArgs a, TmArgs tm, TgArgs tg) {
    __shared__ __attribute__((aligned(8192))) uint32_t planes[2][2][8 * 64];
    __shared__ uint32_t nc_sh[2][64];
    __shared__ __attribute__((aligned(16))) uint16_t xts[TMW_WAVES][TM_GROUP][8][64];      // the waves' transposition tiles
    __shared__ __attribute__((aligned(16))) tg_u32x4 tmask[2][9];                         // [0][x]: the first x half-words of a chunk; [1][x]: all but them
    __shared__ WaveBook books[TMW_WAVES];
    __shared__ uint32_t s_ck;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* pl = &planes[0][0][0];
    WaveBook& book = books[wv];
    book_init(book, lane);
    if (lane == 0) book.src = 1;
    if (threadIdx.x < 18) {
        const uint32_t x = threadIdx.x % 9u, inv = threadIdx.x >= 9 ? 0xffffffffu : 0u;
        uint32_t w[4];
        for (uint32_t d = 0; d < 4; ++d) w[d] = ((2u * d < x ? 0xffffu : 0u) | (2u * d + 1u < x ? 0xffff0000u : 0u)) ^ inv;
        tmask[threadIdx.x / 9u][x] = tg_u32x4{w[0], w[1], w[2], w[3]};
    }
    const uint32_t thr = bq_threshold(a), pkl0 = lds_addr(pl + lane);
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));
    const uint32_t nchunks = rl(*tg.nchunks, 0);
    TgStat stat; stat.ev = 0; stat.sg = 0; stat.ne = 0;
    for (bool first = true;; first = false) {
        __syncthreads();
        if (threadIdx.x == 0) s_ck = first ? blockIdx.x : (uint32_t)atomicAdd(&a.scalars[SC_QWALK], 1ull) + gridDim.x;
        __syncthreads();
        const uint32_t ck = rl(s_ck, 0);
        if (ck >= nchunks) break;
        const uint32_t jx_end = rl(tm.chunk_start[ck + 1], 0), jx0 = rl(tm.chunk_start[ck], 0);
        if (jx0 >= jx_end) continue;
        // A job's record holds everything its workgroup needs (lsg_ctx.h TmJob); the NEXT job's record is requested when this one's has
        // been read, and the keys of the next job's first group when this job's entries are done - a job starts with its data on the way
        // instead of four dependent round trips (chunk -> job -> tile offsets -> keys)
        uint32_t jw = 0;
        if (lane < TM_JOB_WORDS) jw = reinterpret_cast<const uint32_t*>(tm.jobs + jx0)[lane];
        uint32_t e0 = rl(jw, 0), e1 = rl(jw, 1), w0 = rl(jw, 2), slab = rl(jw, 3), nj = rl(jw, 4), tcnt = rl(jw, 5), tile = rl(jw, 6), emid = rl(jw, 7), base = rl(jw, 8), off = rl(jw, 9);
        int32_t tstart = (int32_t)rl(jw, 10); int tid = (int)rl(jw, 11);
        TgKeys K0 = tg_load_keys(tg, ((wv ? emid : e0) >> 3) * 8, base, off, tcnt, lane);
        for (uint32_t jx = jx0; jx < jx_end; ++jx) {
            {   // (every lane loads - a clamped place past the chunk's last job - so that nothing here has to wait)
                const uint32_t jn = jx + 1 < jx_end ? jx + 1 : jx;
                jw = reinterpret_cast<const uint32_t*>(tm.jobs + jn)[lane < TM_JOB_WORDS ? lane : 0];
            }
            const bool in_region = tile >= a.tile_lo && tile < a.tile_hi, counting = in_region && !(nj & TMJ_WIDE);
            int refb = 'N';
            if (nj == 1 && counting) { const int64_t pos = (int64_t)tstart + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            __syncthreads();                                   // both waves are done with the job before
            if (counting) {
#pragma unroll
                for (int i = 0; i < 2 * 2 * 8 * 64 / (4 * TMW_WAVES * 64); ++i) reinterpret_cast<uint4*>(pl)[i * (TMW_WAVES * 64) + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
                (&nc_sh[0][0])[threadIdx.x] = 0;
            }
            __syncthreads();
            TmState st; st.nc = 0; st.mask = 0;
            const uint32_t s0r = wv ? emid : e0, s1r = wv ? e1 : emid;
            if (s1r > s0r) {
                tg_range(a, tm, tg, st, stat, s0r, s1r, base, off, tcnt, in_region, counting, thr, pkl0, one, lane, xts[wv], tmask, K0);
                if (counting) {
                    if (st.nc & 0xffffu) atomicAdd(&nc_sh[0][lane], st.nc & 0xffffu);
                    if (st.nc >> 16) atomicAdd(&nc_sh[1][lane], st.nc >> 16);
                }
            }
            // this job's fields for the emission below; then the next job's (its record has arrived long ago) and its first keys
            const uint32_t c_w0 = w0, c_slab = slab, c_nj = nj, c_tcnt = tcnt; const int32_t c_tstart = tstart; const int c_tid = tid;
            e0 = rl(jw, 0); e1 = rl(jw, 1); w0 = rl(jw, 2); slab = rl(jw, 3); nj = rl(jw, 4); tcnt = rl(jw, 5); tile = rl(jw, 6); emid = rl(jw, 7); base = rl(jw, 8); off = rl(jw, 9);
            tstart = (int32_t)rl(jw, 10); tid = (int)rl(jw, 11);
            K0 = tg_load_keys(tg, ((wv ? emid : e0) >> 3) * 8, base, off, tcnt, lane);
            __syncthreads();
            const int ct = tm.ct_base + wv;
            if (counting && ct < a.n_ct) {                   // the tile's units of this pass: wave = cell type (as in k_tm_walk)
                const uint32_t* pc = pl + wv * 1024;
                uint32_t dp = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp += pc[512 + k * 64 + lane] & 0xffffu;
                const TmCounters tot{pc, lane, dp - nc_sh[wv][lane]};
                if (c_nj == 1) {
                    if (c_tcnt <= 256u) emit_unit<TmCounters, true>(a, tot, c_w0 + ct, ct, c_tid, c_tstart, lane, &book, false, refb, 0);
                    else emit_unit<TmCounters, false>(a, tot, c_w0 + ct, ct, c_tid, c_tstart, lane, &book, false, refb, 1);
                } else {
                    uint32_t* dst = a.macc + (uint64_t)(c_slab + (uint32_t)ct * c_nj) * (NCTR * 64);
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
    {   // admitted events / segments / entries of this workgroup (k_resolve_stats adds the slots up)
        unsigned long long ev = stat.ev, sg = stat.sg, ne = stat.ne;
        for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
        if (lane == 0 && ne) {
            unsigned long long* slot = tg.stat_slots + (size_t)((blockIdx.x * TMW_WAVES + wv) % IX_STAT_SLOTS) * 8;
            atomicAdd(&slot[0], ev); atomicAdd(&slot[1], sg); atomicAdd(&slot[2], ne);
        }
    }
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

// ================================================================================================
// The load's count WITHOUT a store (lsg_set_store_policy(LSG_STORE_SKIP_WHEN_COUNTED)): a rule that counts a BAM once and never again
// needs no resident form of the events - the 24 GB of transposed blocks k_tm_gather_count writes at C2 are as much traffic as the
// events themselves.  Same plan, jobs, planes, run logic and rows; a wave reads an entry's events where the caller left them:
// lane = position of the tile, ONE 2-byte buffer load per entry through a descriptor of its own (base = the entry's first event,
// num_records = its events' bytes, offset = 2 (lane - first position): the lanes outside the entry get zeros without a memory request,
// and nothing beside the entry's own events is ever fetched - no edge cases at the ends of the array).  Per group of entries:
//   K  the sorted keys and values, two groups ahead
//   W  per lane: the entry's address | bytes | first position (two words, read lane by lane into SGPRs when the entry's load is
//      issued), and what its meta word needs (admission bit of its read, cell type of its barcode: issued here, read a group later)
//   E  the events, sixteen entries to a batch of loads
// (an event is SIGN-extended into its register: the compiler keeps that out of the loop, where a zero-extension of the 2-byte load's
// value was an operation per entry that waits for the load wherever it is placed; tm_add<false> reads the register's low half only)
struct TdW { uint32_t lo, hi; };     // hi: address bits 32..47 | 2 * events << 16 | 2 * first position << 24
// 64 entries to a group, one per lane: what a group costs beside its entries (the keys' way into words, the meta words, the run verdicts:
// ~130 vector operations) is paid once per 64 entries (32 to a group, with the neighbours' keys in two spare lanes: 8.65 ms at C2; 64: 8.4).
// The barcode BEFORE the group travels as a scalar from the group before (the range's first one: a load of its own), the one AFTER it is
// lane 0 of the next group's keys, which have arrived when the group's meta words are finished (an iteration after its words were made).
// Four quarters of 16 entries alternate between two sets of registers: 16 to 32 loads are in flight while 16 entries are counted.
// The fences pin the ORDER the loads are issued in - look-ups, keys, quarter by quarter - in the prologue as in the loop: the counter of
// loads in flight is in-order, and where two orders meet at the loop's head the compiler waits for everything.
#ifndef LSG_TD_Q
#define LSG_TD_Q 16
#endif
#ifndef LSG_TD_WAVES
#define LSG_TD_WAVES 6
#endif
#ifndef LSG_TW_WAVES
#define LSG_TW_WAVES 5
#endif
constexpr int TD_Q = LSG_TD_Q, TD_NQ = 64 / TD_Q;      // entries to a batch of loads (two batches of registers alternate), batches to a group.
// (8 to a batch fit 64 registers and 8 waves per SIMD, 16 workgroups per CU: 8.1-8.8 ms, no better than 16 to a batch at 6 waves: 8.0-8.2)
struct TdKeys64 { uint64_t k; uint32_t v; };
// KO: the sort carried keys alone (store.hip build_store, keys_only): forward strand and first-of-segment are bits 63 and 62 of the key,
// where RV_FWD and RV_SEGFIRST sit in the upper word; no read index (every stored read is admitted: run_gather_count checks)
template <bool KO>
__device__ __forceinline__ TdKeys64 td_load_keys64(const TgArgs& tg, uint32_t i_first, uint32_t off, uint32_t n, int lane) {
    TdKeys64 r;
    const uint32_t i = i_first + (uint32_t)lane, ic = i < n ? i : n - 1u;       // (past the tile's end: its last entry, which is never used)
    r.k = __builtin_nontemporal_load(tg.key + off + ic);
    if (KO) r.v = (uint32_t)(r.k >> 32) & (TG_RV_FWD | TG_RV_SEGFIRST);
    else r.v = __builtin_nontemporal_load(tg.rdv + off + ic);
    return r;
}
template <bool KO, bool AL>
__device__ __forceinline__ void td_range64(const CountArgs& a, const TmArgs& tm, const TgArgs& tg, TmState& st, TgStat& stat, uint32_t i0, uint32_t i1, uint32_t off, uint32_t n,
                                           uint32_t thr, uint32_t pkl0, uint32_t one, int lane, TdKeys64 K, uint64_t key_before) {
    const int ng = (int)((i1 - i0 + 63u) >> 6);
    const uint32_t cbm = (1u << tg.cb_bits) - 1u;
    const uint32_t lane2 = 2u * (uint32_t)lane;
    const uint64_t evb = (uint64_t)(uintptr_t)tg.events;
    const uint32_t* admp = a.adm ? reinterpret_cast<const uint32_t*>(a.adm) : reinterpret_cast<const uint32_t*>(a.celltype_of);
    const uint32_t adm_all = a.adm ? 0u : 0xffffffffu;
    uint32_t cb_carry = rl((uint32_t)key_before & cbm, 0);             // barcode of the entry before the group (i0 == 0: not looked at)
    uint32_t cb_last = 0;                                              // barcode of the last entry of the group whose meta words are pending
    auto words = [&](int g, const TdKeys64& K, TdW& w, TgPre& pre) {
        const uint32_t i = i0 + 64u * (uint32_t)g + (uint32_t)lane;
        const bool valid = i < i1;
        const uint32_t cb = (uint32_t)K.k & cbm;
        const uint32_t up = (uint32_t)__shfl_up((int)cb, 1), dn = (uint32_t)__shfl_down((int)cb, 1);
        const uint32_t cb_prev = lane == 0 ? cb_carry : up;
        const uint32_t geom = (uint32_t)(K.k >> tg.cb_bits), first = geom & 63u, nev = valid ? ((geom >> 6) & 63u) + 1u : 0u;
        const uint64_t fld = KO ? (K.k >> (tg.cb_bits + 12)) & tg.src_mask : K.k >> (tg.cb_bits + 12);
        // AL: the address of the entry's LINE: every lane loads its two bytes of it (one request, inside the array: a line that holds an
        // event of the array lies in that event's page), the lanes outside the entry are left out by tm_add<.., true>
        const uint64_t addr = AL ? evb + (fld << 7) : evb + 2ull * ((fld << tg.src_shift) | (tg.src_shift ? first : 0u));
        const bool rs = i == 0 || cb_prev != cb;
        const bool nd = i + 1 == n || (lane < 63 && dn != cb);           // the next entry is another barcode's (lane 63: decided when the next keys are there)
        w.lo = (uint32_t)addr;
        w.hi = AL ? (uint32_t)(addr >> 32) : ((uint32_t)(addr >> 32) & 0xffffu) | (nev << 17) | (first << 25);
        const uint32_t r = K.v & TG_RV_READ;
        pre.ctv = reinterpret_cast<const uint32_t*>(a.celltype_of)[(cb < (uint32_t)a.n_cb ? cb : 0u) >> 2];
        pre.admw = admp[a.adm && r < (uint32_t)a.n_reads ? r >> 5 : 0u] | adm_all;
        pre.bits = (valid ? 1u : 0u) | (cb < (uint32_t)a.n_cb ? 2u : 0u) | ((K.v & TG_RV_FWD) ? 4u : 0u) | (nd ? 8u : 0u) | (rs ? 16u : 0u) |
                   ((r & 31u) << 8) | (nev << 16) | ((K.v & TG_RV_SEGFIRST) ? (1u << 24) : 0u) | ((cb & 3u) << 26);
        if (AL) pre.bits |= ((first & 7u) << 5) | ((first >> 3) << 13);      // (the meta word's low six bits, finish_meta)
        cb_last = cb_carry = rl(cb, 63);
    };
    // cb_after: barcode of the entry after the group (the next group's lane 0)
    auto finish_meta = [&](const TgPre& pre, uint32_t cb_of_last, uint32_t cb_after) -> uint32_t {
        if (!(pre.bits & 1u)) return TMM_SKIP;
        uint32_t cls = 2;
        const bool ok = (pre.bits & 2u) != 0 && ((pre.admw >> ((pre.bits >> 8) & 31u)) & 1u);
        if (ok) { const uint32_t ct = (pre.ctv >> (((pre.bits >> 26) & 3u) * 8u)) & 0xffu; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
        if (cls < 2) { stat.ev += (pre.bits >> 16) & 0xffu; stat.sg += (pre.bits >> 24) & 1u; ++stat.ne; }
        const bool nd = (pre.bits & 8u) != 0 || (lane == 63 && cb_after != cb_of_last);
        const bool single = (pre.bits & 16u) != 0 && nd;
        uint32_t M = cls < 2 ? (cls ? ((AL ? TMM_CT10 : TMM_CT4) | TMM_CT12) : 0u) | ((pre.bits & 4u) ? TMM_FWD : 0u) | (single ? TMM_SINGLE : 0u) : TMM_SKIP;
        if (AL && cls < 2) {                                                                  // the entry's lanes: first, 64 - end
            const uint32_t first = ((pre.bits >> 5) & 7u) | (((pre.bits >> 13) & 7u) << 3), nev = (pre.bits >> 16) & 0xffu;
            M |= first | (((64u - first - nev) & 63u) << 13);
        }
        if (pre.bits & 16u) M |= TMM_RS;
        return M;
    };
    uint32_t open_in = 0;
    auto verdicts = [&](uint32_t M) -> uint32_t {                           // (tm_walk_range's, over 64 entries)
        const bool there = !(M & TMM_SKIP), multi = there && !(M & TMM_SINGLE), rs = (M & TMM_RS) != 0;
        const unsigned long long A = __ballot(multi), R = __ballot(rs);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long rb = R & below;
        const unsigned long long seg = rb ? below & ~((1ull << (63 - __clzll((long long)rb))) - 1ull) : below;
        const bool ob = (A & seg) != 0ull || (!rb && open_in);
        if (rs && ob) M |= TMM_CLOSE;
        if (multi && (rs || !ob)) M |= TMM_FIRST;
        const unsigned long long segl = R ? ~((1ull << (63 - __clzll((long long)R))) - 1ull) : ~0ull;
        open_in = ((A & segl) != 0ull || (!R && open_in)) ? 1u : 0u;
        return M;
    };
    auto issue = [&](const TdW& w, int q, uint32_t (&E)[TD_Q]) {
#pragma unroll
        for (int u = 0; u < TD_Q; ++u) {
            const uint32_t lo = rl(w.lo, q * TD_Q + u), hi = rl(w.hi, q * TD_Q + u);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)(((uint64_t)(hi & 0xffffu) << 32) | lo)), 0, (int)((hi >> 16) & 0xffu), 0x00020000);
            E[u] = (uint32_t)(int32_t)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(lane2 - (hi >> 24)), 0, 0);
        }
    };
    // AL: a scalar base per entry, the lane's 32-bit offset the same for all - `global_load_sshort v, v_off, s[base]`, which the compiler
    // selects only while it sees the offset's zero-extension beside the load (hoisted out of the loop as a 64-bit pair it becomes a
    // 64-bit vector add per entry: the offset is made opaque once per batch)
    auto issue_line = [&](const TdW& w, int q, uint32_t (&E)[TD_Q]) {
        uint32_t off = lane2;
        asm volatile("" : "+v"(off));
        // (eight bases first, then their eight loads: a base fresh from v_readlane needs five wait states before a load may read it)
#pragma unroll
        for (int u0 = 0; u0 < TD_Q; u0 += 8) {
            uint64_t base[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) base[u] = ((uint64_t)rl(w.hi, q * TD_Q + u0 + u) << 32) | rl(w.lo, q * TD_Q + u0 + u);
            asm volatile("" : "+s"(base[0]), "+s"(base[1]), "+s"(base[2]), "+s"(base[3]), "+s"(base[4]), "+s"(base[5]), "+s"(base[6]), "+s"(base[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u)
                E[u0 + u] = (uint32_t)(int32_t)*(const __attribute__((address_space(1))) int16_t*)((const __attribute__((address_space(1))) char*)(uintptr_t)base[u] + (uint64_t)off);
        }
    };
    auto consume = [&](const uint32_t (&E)[TD_Q], int q, uint32_t M) {
#pragma unroll
        for (int u = 0; u < TD_Q; ++u) tm_add<false, AL>(st, rl(M, q * TD_Q + u), E[u], thr, pkl0, one);
    };
    auto fence = []() { asm volatile("" ::: "memory"); };
    uint32_t EA[TD_Q], EB[TD_Q];
    TdW wc, wn; TgPre pre;
    words(0, K, wc, pre);
    fence();
    K = td_load_keys64<KO>(tg, i0 + 64u, off, n, lane);
    fence();
    if (AL) issue_line(wc, 0, EA); else issue(wc, 0, EA);
    fence();
    if (AL) issue_line(wc, 1, EB); else issue(wc, 1, EB);
    fence();
    for (int g = 0; g < ng; ++g) {
        const uint32_t cb_after = rl((uint32_t)K.k & cbm, 0);               // (K: the keys of group g + 1, asked for an iteration ago)
        const uint32_t Mc = verdicts(finish_meta(pre, cb_last, cb_after));
        words(g + 1, K, wn, pre);                                           // (past the range's last group: entries that are not there)
        fence();
        K = td_load_keys64<KO>(tg, i0 + 64u * (uint32_t)(g + 2), off, n, lane);
        fence();
#pragma unroll
        for (int sb = 0; sb < TD_NQ; sb += 2) {                // batch sb is counted while sb + 1 is in flight; its registers take batch sb + 2's loads
            consume(EA, sb, Mc);
            if (AL) issue_line(sb + 2 < TD_NQ ? wc : wn, (sb + 2) % TD_NQ, EA); else issue(sb + 2 < TD_NQ ? wc : wn, (sb + 2) % TD_NQ, EA);
            consume(EB, sb + 1, Mc);
            if (AL) issue_line(sb + 3 < TD_NQ ? wc : wn, (sb + 3) % TD_NQ, EB); else issue(sb + 3 < TD_NQ ? wc : wn, (sb + 3) % TD_NQ, EB);
        }
        wc = wn;
    }
    st.nc += st.mask & 0x10001u; st.mask = 0;
}

template <bool KO, bool AL>
__global__ __launch_bounds__(TMW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(LSG_TD_WAVES))) void k_tm_count_direct(CountArgs a, TmArgs tm, TgArgs tg) {
    __shared__ __attribute__((aligned(8192))) uint32_t planes[2][2][8 * 64];
    __shared__ uint32_t nc_sh[2][64];
    __shared__ WaveBook books[TMW_WAVES];
    __shared__ uint32_t s_q[4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* pl = &planes[0][0][0];
    WaveBook& book = books[wv];
    book_init(book, lane);
    if (lane == 0) book.src = 1;
    const uint32_t thr = AL ? (uint32_t)a.min_bq : bq_threshold(a), pkl0 = lds_addr(pl + lane);      // (AL: 1 <= min_bq <= 255, run_gather_count)
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));
    TgStat stat; stat.ev = 0; stat.sg = 0; stat.ne = 0;
    // WHO takes WHICH job.  The jobs lie in tile order, and a segment's entries in two adjacent tiles are adjacent in the caller's array: the
    // 128-byte line that holds the end of the one holds the start of the other, and the fabric moves 1.77 lines per entry (44.9 GB a launch
    // for 19.9 GB of events and keys) where 0.8 would do if every line were fetched once.  The jobs are dealt in blocks of 64 consecutive
    // ones to the eight XCDs, and the workgroups of an XCD take the jobs of its blocks ONE BY ONE off the XCD's queue, so that at any moment
    // an XCD works on a few hundred consecutive tiles, started a fraction of a microsecond apart, whose shared lines could meet in its L2.
    // Measured: they mostly do not (42.1 GB; 12 % of the L2's read requests hit) - a job lasts ~50 us, the two tiles' runs through the
    // same barcodes drift apart by more than the ~6 us a line stays in 4 MB of L2 - but the finer hand-out balances the tail better than
    // chunks did (8.4 -> 8.2 ms with twelve workgroups per CU), so it stays.  A workgroup whose XCD has run dry helps the next one.  The id
    // of the job after next is asked for (one atomic of thread 0) while this job is counted; the next job's record and first keys travel
    // as before.
    const uint32_t njobs = tm.njobs, nblocks = (njobs + 63u) >> 6;
    const uint32_t xcd = blockIdx.x & 7u;      // (workgroups are dealt to the XCDs round-robin)
    constexpr uint32_t NO_JOB = 0xffffffffu;
    uint32_t steal = 0;                                                                   // (thread 0's: queues it has found empty)
    auto job_of = [&](uint32_t q, uint32_t n) -> uint32_t {                               // n-th job of XCD q's blocks, or NO_JOB past them
        const uint32_t blk = 8u * (n >> 6) + q;
        const uint32_t jx = (blk << 6) + (n & 63u);
        return blk < nblocks && jx < njobs ? jx : NO_JOB;
    };
    auto take = [&]() -> uint32_t {                                                       // (thread 0)
        while (steal < 8u) {
            const uint32_t q = (xcd + steal) & 7u;
            const uint32_t n = (uint32_t)atomicAdd(tg.queues + q * 16u, 1ull);
            const uint32_t jx = job_of(q, n);
            if (jx != NO_JOB) return jx;
            if (8u * (n >> 6) + q >= nblocks) ++steal;                                    // (past the queue's last block; a short last block: ask again)
        }
        return NO_JOB;
    };
    if (threadIdx.x == 0) { s_q[0] = take(); s_q[1] = take(); }
    __syncthreads();
    uint32_t cur = rl(s_q[0], 0), nxt = rl(s_q[1], 0);
    if (cur != NO_JOB) {
        uint32_t jw = 0;
        if (lane < TM_JOB_WORDS) jw = reinterpret_cast<const uint32_t*>(tm.jobs + cur)[lane];
        uint32_t e0 = rl(jw, 0), e1 = rl(jw, 1), w0 = rl(jw, 2), slab = rl(jw, 3), nj = rl(jw, 4), tcnt = rl(jw, 5), tile = rl(jw, 6), emid = rl(jw, 7), base = rl(jw, 8), off = rl(jw, 9);
        int32_t tstart = (int32_t)rl(jw, 10); int tid = (int)rl(jw, 11);
        // (the first group's keys of the wave's range, and the entry before it)
        auto first_i = [&]() -> uint32_t { return (wv ? emid : e0) - base; };
        TdKeys64 K64{}; uint64_t kb = 0;
        auto prefetch = [&]() { const uint32_t i = first_i(); K64 = td_load_keys64<KO>(tg, i, off, tcnt, lane); kb = __builtin_nontemporal_load(tg.key + off + (i ? i - 1u : 0u)); };
        prefetch();
        while (true) {
            jw = reinterpret_cast<const uint32_t*>(tm.jobs + (nxt != NO_JOB ? nxt : cur))[lane < TM_JOB_WORDS ? lane : 0];      // (every lane loads: nothing here waits)
            const bool counting = tile >= a.tile_lo && tile < a.tile_hi && !(nj & TMJ_WIDE);      // (the wide jobs: k_tm_walk_wide_direct)
            int refb = 'N';
            if (nj == 1 && counting) { const int64_t pos = (int64_t)tstart + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            __syncthreads();                                   // both waves are done with the job before
            // the job after next: the fast way is ONE atomic whose answer is looked at after this job's entries (a queue that has run dry is rare)
            uint32_t q_n = 0;
            const uint32_t q_q = (xcd + steal) & 7u;
            if (threadIdx.x == 0 && steal < 8u) q_n = (uint32_t)atomicAdd(tg.queues + q_q * 16u, 1ull);
            if (counting) {
#pragma unroll
                for (int i = 0; i < 2 * 2 * 8 * 64 / (4 * TMW_WAVES * 64); ++i) reinterpret_cast<uint4*>(pl)[i * (TMW_WAVES * 64) + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
                (&nc_sh[0][0])[threadIdx.x] = 0;
            }
            __syncthreads();
            // the tile's entries [i0, i1) of this wave, as indices into the tile's part of the sort's output (the plan cuts jobs in padded
            // store indices: base is the tile's first one)
            const uint32_t s0r = wv ? emid : e0, s1r = wv ? e1 : emid;
            const uint32_t i0 = s0r - base, i1 = s1r - base < tcnt ? s1r - base : tcnt;
            if (counting && i1 > i0) {
                TmState st; st.nc = 0; st.mask = 0;
                td_range64<KO, AL>(a, tm, tg, st, stat, i0, i1, off, tcnt, thr, pkl0, one, lane, K64, kb);
                if (st.nc & 0xffffu) atomicAdd(&nc_sh[0][lane], st.nc & 0xffffu);
                if (st.nc >> 16) atomicAdd(&nc_sh[1][lane], st.nc >> 16);
            }
            if (threadIdx.x == 0) {
                uint32_t jn2 = NO_JOB;
                if (steal < 8u) {
                    jn2 = job_of(q_q, q_n);
                    if (jn2 == NO_JOB) { if (8u * (q_n >> 6) + q_q >= nblocks) ++steal; jn2 = take(); }
                }
                s_q[2] = jn2;
            }
            const uint32_t c_w0 = w0, c_slab = slab, c_nj = nj, c_tcnt = tcnt; const int32_t c_tstart = tstart; const int c_tid = tid;
            e0 = rl(jw, 0); e1 = rl(jw, 1); w0 = rl(jw, 2); slab = rl(jw, 3); nj = rl(jw, 4); tcnt = rl(jw, 5); tile = rl(jw, 6); emid = rl(jw, 7); base = rl(jw, 8); off = rl(jw, 9);
            tstart = (int32_t)rl(jw, 10); tid = (int)rl(jw, 11);
            prefetch();
            __syncthreads();
            const uint32_t nn = rl(s_q[2], 0);
            const int ct = tm.ct_base + wv;
            if (counting && ct < a.n_ct) {                   // the tile's units of this pass: wave = cell type (as in k_tm_walk)
                const uint32_t* pc = pl + wv * 1024;
                uint32_t dp = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp += pc[512 + k * 64 + lane] & 0xffffu;
                const TmCounters tot{pc, lane, dp - nc_sh[wv][lane]};
                if (c_nj == 1) {
                    if (c_tcnt <= 256u) emit_unit<TmCounters, true>(a, tot, c_w0 + ct, ct, c_tid, c_tstart, lane, &book, false, refb, 0);
                    else emit_unit<TmCounters, false>(a, tot, c_w0 + ct, ct, c_tid, c_tstart, lane, &book, false, refb, 1);
                } else {
                    uint32_t* dst = a.macc + (uint64_t)(c_slab + (uint32_t)ct * c_nj) * (NCTR * 64);
                    dst[lane] = tot.NCDUP();
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t lo = pc[k * 64 + lane], hi = pc[512 + k * 64 + lane];
                        dst[(1 + k) * 64 + lane] = hi >> 16; dst[(9 + k) * 64 + lane] = hi & 0xffffu;
                        dst[(17 + k) * 64 + lane] = lo & 0xfffffu; dst[(25 + k) * 64 + lane] = lo >> 20;
                    }
                }
            }
            if (nxt == NO_JOB) break;
            cur = nxt; nxt = nn;
        }
    }
    lds_fence();
    __syncthreads();
    {
        unsigned long long ev = stat.ev, sg = stat.sg, ne = stat.ne;
        for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
        if (lane == 0 && ne) {
            unsigned long long* slot = tg.stat_slots + (size_t)((blockIdx.x * TMW_WAVES + wv) % IX_STAT_SLOTS) * 8;
            atomicAdd(&slot[0], ev); atomicAdd(&slot[1], sg); atomicAdd(&slot[2], ne);
        }
    }
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

// One entry of a 128-position WINDOW at the lane's two positions (k_tm_count_win below): what two tm_add<.., true> calls do - the lane's
// even position from the low half-word of ev, its odd position from the high one, each with a run state of its own - with ONE pass
// through the entry's scalar decisions (close / not there / run of one / first of a run / further entry of a run: the same for both
// positions) and ONE 64-bit LDS atomic per position instead of two 32-bit ones (an LDS instruction costs a CU ~4.3 cycles whatever its
// width up to 32 bits, 6.3 at 64: tools/lds_rate.hip).  A counter is 8 bytes: low word quality sum [0..19] | forward count [20..31], high word
// count [0..12] | duplicates x 2^symbol [13..31] (a duplicate of symbol class c adds the run's seen-bit of that symbol, 1 << (8 + c), shifted
// up by 5 - no normalising: a counter holds one symbol only, the emit shifts it back; <= 4095 entries a job: 12 + 7 bits);
// byte address = parity << 13 | cell type << 12 | tile << 11 | symbol << 8 | (position >> 1) << 3.
// m: the entry's meta word - TMM_CLOSE, TMM_FIRST, TMM_SKIP, TMM_SINGLE, TMM_CT12, TMM_FWD where tm_add has them, and in the bits between them
// the lanes of the entry's positions: even ones [m & 63, 64 - (m >> 6 & 63)), odd ones [m >> 13 & 63, 64 - (m >> 21 & 63)) (tw_ranges).  The atomics' data are register PAIRS, which inline
// asm can only name by number: v[92:93] = {low word, 1} (v93 holds 1 throughout: `one`), v[94:95] = {low word, 1 | seen << 16}.
__device__ __forceinline__ void tw_add(TmState& s0, TmState& s1, uint32_t m, uint32_t ev, uint32_t thr, uint32_t pk0, uint32_t pk1, uint32_t one) {
    const uint32_t r = m;      // (ONE word: the decisions' bits and the lanes' ranges, TWM_* below)
    if (!TM_ASM) {                                   // the same in plain C++
        const uint32_t ln = threadIdx.x & 63u;
        auto half = [&](TmState& s, uint32_t e, uint32_t f, uint32_t inv, uint32_t pk) {
            if (m & TMM_CLOSE) { s.nc += s.mask & 0x10001u; s.mask = 0; }
            if (m & TMM_SKIP) return;
            const bool counted = (e & 0xffu) >= thr && ln >= f && ln < 64u - inv;
            if (!counted) return;
            const uint32_t addr = (pk | (e & 0x700u)) + (m & TMM_CT12);
            const uint32_t sym8 = (e >> 8) & 15u, ctone = 1u << ((m >> 8) & 16u);
            uint32_t hi = 1u;
            if (m & TMM_SINGLE) s.nc += ctone;
            else if (m & TMM_FIRST) s.mask = (1u << sym8) | ctone;
            else { hi |= (s.mask & (1u << sym8)) << 5; s.mask |= (1u << sym8) | ctone; }
            __hip_atomic_fetch_add((LSG_AS3 unsigned long long*)(uintptr_t)addr, ((unsigned long long)hi << 32) | ((e & 0xffu) | (m & TMM_FWD)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        half(s0, ev & 0xffffu, r & 63u, (r >> 6) & 63u, pk0);
        half(s1, ev >> 16, (r >> 13) & 63u, (r >> 21) & 63u, pk1);
        return;
    }
    uint32_t t0, t1, addr, dlo, elo, ehi, sa, sb, sc, st; unsigned long long xm;
    // the lanes of a half's positions into EXEC (whole: nothing of the half before is left in it)
#define TW_RANGE0 "s_lshr_b32 %[st], %[r], 6\n\ts_lshl_b64 %[xm], -1, %[r]\n\ts_lshr_b64 exec, -1, %[st]\n\ts_and_b64 exec, exec, %[xm]\n\t"
#define TW_RANGE1 "s_lshr_b32 %[st], %[r], 13\n\ts_lshl_b64 %[xm], -1, %[st]\n\ts_lshr_b32 %[st], %[r], 21\n\ts_lshr_b64 exec, -1, %[st]\n\ts_and_b64 exec, exec, %[xm]\n\t"
    // address of the position's counter, low word of the atomic's data (into DLO), EXEC = the lanes that count this event
#define TW_HEAD(WSEL, BSEL, PK, DLO)                                                                                                 \
        "v_and_b32_sdwa %[addr], %[c700], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" WSEL "\n\t"           \
        "v_or3_b32 %[addr], %[addr], " PK ", %[sa]\n\t"                                                                             \
        "v_or_b32_sdwa " DLO ", %[sb], %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BSEL "\n\t"              \
        "v_cmpx_le_u32_sdwa vcc, %[thr], %[ev] src0_sel:DWORD src1_sel:" BSEL "\n\t"
    asm volatile(
        "s_bitcmp1_b32 %[m], 27\n\t"
        "s_cbranch_scc0 1f\n\t"
        "v_and_b32 %[t0], 0x10001, %[mask0]\n\t"                    /* close the run of several entries before this one, both positions */
        "v_and_b32 %[t1], 0x10001, %[mask1]\n\t"
        "v_add_u32 %[nc0], %[nc0], %[t0]\n\t"
        "v_add_u32 %[nc1], %[nc1], %[t1]\n\t"
        "v_mov_b32 %[mask0], 0\n\t"
        "v_mov_b32 %[mask1], 0\n"
        "1:\n\t"
        "s_bitcmp1_b32 %[m], 29\n\t"
        "s_cbranch_scc1 9f\n\t"                                     /* not counted / not there */
        "s_and_b32 %[sa], %[m], 0x1000\n\t"                         /* the cell type's counters */
        "s_and_b32 %[sb], %[m], 0x100000\n\t"                       /* forward strand */
        "s_lshr_b32 %[sc], %[sa], 8\n\t"
        "s_lshl_b32 %[sc], 1, %[sc]\n\t"                            /* bit 0 or bit 16: the cell type's run counter */
        "s_bitcmp1_b32 %[m], 30\n\t"
        "s_cbranch_scc0 2f\n\t"
        TW_RANGE0                                                   /* ---- a run of one entry */
        TW_HEAD("WORD_0", "BYTE_0", "%[pk0]", "v92")
        "ds_add_u64 %[addr], v[92:93]\n\t"
        "v_add_u32 %[nc0], %[sc], %[nc0]\n\t"
        TW_RANGE1
        TW_HEAD("WORD_1", "BYTE_2", "%[pk1]", "v92")
        "ds_add_u64 %[addr], v[92:93]\n\t"
        "v_add_u32 %[nc1], %[sc], %[nc1]\n\t"
        "s_branch 8f\n"
        "2:\n\t"
        "s_bitcmp1_b32 %[m], 28\n\t"
        "s_cbranch_scc0 3f\n\t"
        TW_RANGE0                                                   /* ---- first entry of a longer run that is there */
        TW_HEAD("WORD_0", "BYTE_0", "%[pk0]", "v92")
        "v_bfe_u32 %[t1], %[ev], 8, 4\n\t"                          /* 8 + class */
        "ds_add_u64 %[addr], v[92:93]\n\t"
        "v_lshl_or_b32 %[mask0], %[one], %[t1], %[sc]\n\t"
        TW_RANGE1
        TW_HEAD("WORD_1", "BYTE_2", "%[pk1]", "v92")
        "v_bfe_u32 %[t1], %[ev], 24, 4\n\t"
        "ds_add_u64 %[addr], v[92:93]\n\t"
        "v_lshl_or_b32 %[mask1], %[one], %[t1], %[sc]\n\t"
        "s_branch 8f\n"
        "3:\n\t"
        TW_RANGE0                                                   /* ---- a further entry of the run: its symbol seen before = a duplicate */
        TW_HEAD("WORD_0", "BYTE_0", "%[pk0]", "v94")
        "v_lshlrev_b32_sdwa %[t1], %[ev], %[one] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"      /* the symbol's seen-bit: 1 << (8 + class) */
        "v_and_b32 %[t0], %[mask0], %[t1]\n\t"
        "v_lshl_or_b32 v95, %[t0], 5, %[one]\n\t"
        "ds_add_u64 %[addr], v[94:95]\n\t"
        "v_or3_b32 %[mask0], %[mask0], %[t1], %[sc]\n\t"
        TW_RANGE1
        TW_HEAD("WORD_1", "BYTE_2", "%[pk1]", "v94")
        "v_lshlrev_b32_sdwa %[t1], %[ev], %[one] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "v_and_b32 %[t0], %[mask1], %[t1]\n\t"
        "v_lshl_or_b32 v95, %[t0], 5, %[one]\n\t"
        "ds_add_u64 %[addr], v[94:95]\n\t"
        "v_or3_b32 %[mask1], %[mask1], %[t1], %[sc]\n"
        "8:\n\t"
        "s_mov_b64 exec, -1\n"
        "9:"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [addr] "=&v"(addr), [dlo] "=&{v92}"(dlo), [elo] "=&{v94}"(elo), [ehi] "=&{v95}"(ehi),
          [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc), [st] "=&s"(st), [xm] "=&s"(xm),
          [mask0] "+v"(s0.mask), [nc0] "+v"(s0.nc), [mask1] "+v"(s1.mask), [nc1] "+v"(s1.nc)
        : [ev] "v"(ev), [m] "s"(m), [r] "s"(m), [thr] "s"(thr), [c700] "s"(0x700u), [one] "{v93}"(one), [pk0] "v"(pk0), [pk1] "v"(pk1)
        : "scc", "vcc", "memory");
#undef TW_RANGE0
#undef TW_RANGE1
#undef TW_HEAD
}

// ================================================================================================
// The same count over entries binned by 128-position WINDOWS (store.hip build_store, wsh = 1: events phased modulo 128, keys alone, no
// store).  An entry = (segment x window), its events one aligned 256-byte block of the caller's array: ONE `global_load_dword` per entry -
// lane l gets the events of window positions 2 l and 2 l + 1, so lanes 0..31 hold tile 2 w and lanes 32..63 tile 2 w + 1 - where the tiles'
// count asks for two 128-byte lines with two instructions.  What that buys is in the memory system (tools/block_rate.hip: scattered
// aligned 256-byte blocks arrive at 23 G/s = 5.9 TB/s, scattered 128-byte lines at 29 G/s = 3.7 TB/s, whatever asks for them) and in the
// load's first half (0.58 x the entries through the scatter and the sort).  The counting body is tw_add (above): the lane's even position
// from the low half-word of the register, its odd position from the high one.  A group is 32 entries, one per lane of either half of the
// wave (both halves make the same words: the group's keys are loaded twice over).  Planes: tw_add's (16 KB a workgroup).
struct TwCounters {      // the 8-byte counters of one (tile, cell type) read by position (emit_unit's lane); pl: their words at parity 0, symbol 0, position pair 0
    const uint32_t* pl; int lane; uint32_t ncdup;
    __device__ __forceinline__ uint32_t lo(int k) const { return pl[(lane & 1) * 2048 + k * 64 + (lane >> 1) * 2]; }
    __device__ __forceinline__ uint32_t hi(int k) const { return pl[(lane & 1) * 2048 + k * 64 + (lane >> 1) * 2 + 1]; }
    __device__ __forceinline__ uint32_t BC(int k) const { return hi(k) & 0x1fffu; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return hi(k) >> (13 + k); }
    __device__ __forceinline__ uint32_t BQ(int k) const { return lo(k) & 0xfffffu; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return lo(k) >> 20; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};
struct TwPre { uint32_t bits, ctv, rng; };
// TW_G: entries to a group - 64 (one per lane) or 32 (both halves of the wave work the same 32 words out: half the unrolled loop, twice the
// words' instructions per entry)
#ifndef LSG_TW_GROUP
#define LSG_TW_GROUP 64
#endif
constexpr int TW_G = LSG_TW_GROUP, TW_NQ = TW_G / TD_Q;
static_assert(TW_G == 32 || TW_G == 64, "a group of the windows' count is 32 or 64 entries");
__device__ __forceinline__ void tw_range(const CountArgs& a, const TmArgs& tm, const TgArgs& tg, TmState& st0, TmState& st1, TgStat& stat, uint32_t i0, uint32_t i1, uint32_t off, uint32_t n,
                                         uint32_t thr, uint32_t pkl0, uint32_t one, int lane, uint64_t K, uint64_t key_before) {
    const int ng = (int)((i1 - i0 + (uint32_t)TW_G - 1u) / (uint32_t)TW_G);
    const uint32_t cbm = (1u << tg.cb_bits) - 1u;
    const int e = lane & (TW_G - 1);                                       // the lane's entry of a group
    const uint32_t lane4 = 4u * (uint32_t)lane;
    const uint32_t pkl1 = pkl0 | 8192u;                                    // (the odd positions' counters)
    const uint64_t evb = (uint64_t)(uintptr_t)tg.events;
    auto load_keys = [&](uint32_t i_first) -> uint64_t {
        const uint32_t i = i_first + (uint32_t)e, ic = i < n ? i : n - 1u;
        return __builtin_nontemporal_load(tg.key + off + ic);
    };
    uint32_t cb_carry = rl((uint32_t)key_before & cbm, 0);
    uint32_t cb_last = 0;
    auto words = [&](int g, uint64_t Kk, TdW& w, TwPre& pre) {
        const uint32_t i = i0 + (uint32_t)TW_G * (uint32_t)g + (uint32_t)e;
        const bool valid = i < i1;
        const uint32_t cb = (uint32_t)Kk & cbm;
        const uint32_t up = (uint32_t)__shfl((int)cb, e ? lane - 1 : lane), dn = (uint32_t)__shfl((int)cb, e < TW_G - 1 ? lane + 1 : lane);
        const uint32_t cb_prev = e == 0 ? cb_carry : up;
        const uint32_t geom = (uint32_t)(Kk >> tg.cb_bits), first = geom & 127u, nev = valid ? ((geom >> 7) & 127u) + 1u : 0u, end = first + nev;
        const uint64_t fld = (Kk >> (tg.cb_bits + 14)) & tg.src_mask;
        const uint64_t addr = evb + (fld << 8);                           // the entry's 256-byte block
        const bool rs = i == 0 || cb_prev != cb;
        const bool nd = i + 1 == n || (e < TW_G - 1 && dn != cb);
        w.lo = (uint32_t)addr; w.hi = (uint32_t)(addr >> 32);
        const uint32_t kv = (uint32_t)(Kk >> 32) & (TG_RV_FWD | TG_RV_SEGFIRST);
        pre.ctv = reinterpret_cast<const uint32_t*>(a.celltype_of)[(cb < (uint32_t)a.n_cb ? cb : 0u) >> 2];
        pre.bits = (valid ? 1u : 0u) | (cb < (uint32_t)a.n_cb ? 2u : 0u) | ((kv & TG_RV_FWD) ? 4u : 0u) | (nd ? 8u : 0u) | (rs ? 16u : 0u) |
                   (nev << 16) | ((kv & TG_RV_SEGFIRST) ? (1u << 24) : 0u) | ((cb & 3u) << 26);
        // the lanes of the entry's positions: even positions 2 j in [first, end) <=> j in [ceil(first / 2), ceil(end / 2)); odd ones
        // 2 j + 1 <=> j in [first / 2, end / 2).  None: the range [63, 1)
        uint32_t f0 = (first + 1u) >> 1, e0 = (end + 1u) >> 1, f1 = first >> 1, e1 = end >> 1;
        if (f0 >= e0) { f0 = 63u; e0 = 1u; }
        if (f1 >= e1) { f1 = 63u; e1 = 1u; }
        pre.rng = f0 | (((64u - e0) & 63u) << 6) | (f1 << 13) | (((64u - e1) & 63u) << 21);      // (tw_add: between the meta word's decision bits)
        cb_last = cb_carry = rl(cb, TW_G - 1);
    };
    auto finish_meta = [&](const TwPre& pre, uint32_t cb_of_last, uint32_t cb_after) -> uint32_t {
        if (!(pre.bits & 1u)) return TMM_SKIP;
        uint32_t cls = 2;
        if (pre.bits & 2u) { const uint32_t ct = (pre.ctv >> (((pre.bits >> 26) & 3u) * 8u)) & 0xffu; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
        if (cls < 2 && lane < TW_G) { stat.ev += (pre.bits >> 16) & 0xffu; stat.sg += (pre.bits >> 24) & 1u; ++stat.ne; }
        const bool nd = (pre.bits & 8u) != 0 || (e == TW_G - 1 && cb_after != cb_of_last);
        const bool single = (pre.bits & 16u) != 0 && nd;
        uint32_t M = cls < 2 ? (cls ? TMM_CT12 : 0u) | ((pre.bits & 4u) ? TMM_FWD : 0u) | (single ? TMM_SINGLE : 0u) | pre.rng : TMM_SKIP;
        if (pre.bits & 16u) M |= TMM_RS;
        return M;
    };
    uint32_t open_in = 0;
    auto verdicts = [&](uint32_t M) -> uint32_t {                           // (td_range64's / tm_walk_range's, over the group's entries)
        const bool there = !(M & TMM_SKIP), multi = there && !(M & TMM_SINGLE), rs = (M & TMM_RS) != 0;
        const unsigned long long gm = TW_G == 64 ? ~0ull : 0xffffffffull;
        const unsigned long long A = __ballot(multi) & gm, R = __ballot(rs) & gm;
        const unsigned long long below = (1ull << e) - 1ull;
        const unsigned long long rb = R & below;
        const unsigned long long seg = rb ? below & ~((1ull << (63 - __clzll((long long)rb))) - 1ull) : below;
        const bool ob = (A & seg) != 0ull || (!rb && open_in);
        if (rs && ob) M |= TMM_CLOSE;
        if (multi && (rs || !ob)) M |= TMM_FIRST;
        const unsigned long long segl = R ? ~((1ull << (63 - __clzll((long long)R))) - 1ull) : ~0ull;
        open_in = ((A & segl & gm) != 0ull || (!R && open_in)) ? 1u : 0u;
        return M;
    };
    auto issue_block = [&](const TdW& w, int q, uint32_t (&E)[TD_Q]) {      // (td_range64's issue_line with four bytes a lane)
        uint32_t o = lane4;
        asm volatile("" : "+v"(o));
#pragma unroll
        for (int u0 = 0; u0 < TD_Q; u0 += 8) {
            uint64_t base[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) base[u] = ((uint64_t)rl(w.hi, q * TD_Q + u0 + u) << 32) | rl(w.lo, q * TD_Q + u0 + u);
            asm volatile("" : "+s"(base[0]), "+s"(base[1]), "+s"(base[2]), "+s"(base[3]), "+s"(base[4]), "+s"(base[5]), "+s"(base[6]), "+s"(base[7]));
#pragma unroll
            for (int u = 0; u < 8; ++u)
                E[u0 + u] = *(const __attribute__((address_space(1))) uint32_t*)((const __attribute__((address_space(1))) char*)(uintptr_t)base[u] + (uint64_t)o);
        }
    };
    auto consume = [&](const uint32_t (&E)[TD_Q], int q, uint32_t M) {
#pragma unroll
        for (int u = 0; u < TD_Q; ++u) tw_add(st0, st1, rl(M, q * TD_Q + u), E[u], thr, pkl0, pkl1, one);
    };
    auto fence = []() { asm volatile("" ::: "memory"); };
    uint32_t EA[TD_Q], EB[TD_Q];
    TdW wc, wn; TwPre pre;
    words(0, K, wc, pre);
    fence();
    K = load_keys(i0 + (uint32_t)TW_G);
    fence();
    issue_block(wc, 0, EA);
    fence();
    issue_block(wc, 1, EB);
    fence();
    for (int g = 0; g < ng; ++g) {
        const uint32_t cb_after = rl((uint32_t)K & cbm, 0);
        const uint32_t Mc = verdicts(finish_meta(pre, cb_last, cb_after));
        words(g + 1, K, wn, pre);
        fence();
        K = load_keys(i0 + (uint32_t)TW_G * (uint32_t)(g + 2));
        fence();
#pragma unroll
        for (int sb = 0; sb < TW_NQ; sb += 2) {                // (td_range64's alternation of the two batches of registers)
            consume(EA, sb, Mc); issue_block(sb + 2 < TW_NQ ? wc : wn, (sb + 2) % TW_NQ, EA);
            consume(EB, sb + 1, Mc); issue_block(sb + 3 < TW_NQ ? wc : wn, (sb + 3) % TW_NQ, EB);
        }
        wc = wn;
    }
    st0.nc += st0.mask & 0x10001u; st0.mask = 0;
    st1.nc += st1.mask & 0x10001u; st1.mask = 0;
}

__global__ __launch_bounds__(TMW_WAVES * 64) __attribute__((amdgpu_waves_per_eu(LSG_TW_WAVES))) void k_tm_count_win(CountArgs a, TmArgs tm, TgArgs tg) {
    __shared__ __attribute__((aligned(16384))) uint32_t planes[2][2][2][8][32][2];    // tw_add's counters: [parity of the position][cell type of the pass][tile][symbol][position >> 1][low, high word]
    __shared__ uint32_t nc_sh[2][2][64];                                              // [cell type][tile][position]: runs that counted an event
    __shared__ WaveBook books[TMW_WAVES];
    __shared__ uint32_t s_q[4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* pl = &planes[0][0][0][0][0][0];
    WaveBook& book = books[wv];
    book_init(book, lane);
    if (lane == 0) book.src = 1;
    const uint32_t thr = (uint32_t)a.min_bq;                                          // (1 <= min_bq <= 255: build_store)
    const uint32_t pkl0 = lds_addr(pl) | ((uint32_t)(lane >> 5) << 11) | ((uint32_t)(lane & 31) << 3);      // (the lane's tile and position pair)
    uint32_t one = 1u;
    asm volatile("" : "+v"(one));
    TgStat stat; stat.ev = 0; stat.sg = 0; stat.ne = 0;
    // (who takes which job: k_tm_count_direct's queues, a job = a window or a run-aligned piece of a deep one)
    const uint32_t njobs = tm.njobs, nblocks = (njobs + 63u) >> 6;
    const uint32_t xcd = blockIdx.x & 7u;
    constexpr uint32_t NO_JOB = 0xffffffffu;
    uint32_t steal = 0;
    auto job_of = [&](uint32_t q, uint32_t nn) -> uint32_t {
        const uint32_t blk = 8u * (nn >> 6) + q;
        const uint32_t jx = (blk << 6) + (nn & 63u);
        return blk < nblocks && jx < njobs ? jx : NO_JOB;
    };
    auto take = [&]() -> uint32_t {
        while (steal < 8u) {
            const uint32_t q = (xcd + steal) & 7u;
            const uint32_t nn = (uint32_t)atomicAdd(tg.queues + q * 16u, 1ull);
            const uint32_t jx = job_of(q, nn);
            if (jx != NO_JOB) return jx;
            if (8u * (nn >> 6) + q >= nblocks) ++steal;
        }
        return NO_JOB;
    };
    if (threadIdx.x == 0) { s_q[0] = take(); s_q[1] = take(); }
    __syncthreads();
    uint32_t cur = rl(s_q[0], 0), nxt = rl(s_q[1], 0);
    if (cur != NO_JOB) {
        uint32_t jw = 0;
        if (lane < TM_JOB_WORDS) jw = reinterpret_cast<const uint32_t*>(tm.jobs + cur)[lane];
        uint32_t e0 = rl(jw, 0), e1 = rl(jw, 1), w0 = rl(jw, 2), slab = rl(jw, 3), nj = rl(jw, 4), tcnt = rl(jw, 5), win = rl(jw, 6), emid = rl(jw, 7), base = rl(jw, 8), off = rl(jw, 9);
        int32_t tstart = (int32_t)rl(jw, 10); int tid = (int)rl(jw, 11);
        auto first_i = [&]() -> uint32_t { return (wv ? emid : e0) - base; };
        uint64_t K64 = 0, kb = 0;
        auto prefetch = [&]() {
            const uint32_t i = first_i(), ie = i + (uint32_t)(lane & (TW_G - 1)), ic = ie < tcnt ? ie : tcnt - 1u;
            K64 = __builtin_nontemporal_load(tg.key + off + ic); kb = __builtin_nontemporal_load(tg.key + off + (i ? i - 1u : 0u));
        };
        prefetch();
        while (true) {
            jw = reinterpret_cast<const uint32_t*>(tm.jobs + (nxt != NO_JOB ? nxt : cur))[lane < TM_JOB_WORDS ? lane : 0];
            // the wave's tile of the window: tile 2 win + wv, whose units this wave finishes
            const uint32_t my_tile = 2u * win + (uint32_t)wv;
            const bool in_a = 2u * win >= a.tile_lo && 2u * win < a.tile_hi, in_b = 2u * win + 1u >= a.tile_lo && 2u * win + 1u < a.tile_hi;
            const bool counting = (in_a || in_b) && !(nj & TMJ_WIDE);
            const bool mine = counting && (wv ? in_b : in_a);
            int refb = 'N';
            if (nj == 1 && mine) { const int64_t pos = (int64_t)tstart + 64 * wv + lane; if (pos >= 1 && pos < a.contig_len[tid]) refb = a.ref_ptr[tid][pos]; }
            __syncthreads();
            uint32_t q_n = 0;
            const uint32_t q_q = (xcd + steal) & 7u;
            if (threadIdx.x == 0 && steal < 8u) q_n = (uint32_t)atomicAdd(tg.queues + q_q * 16u, 1ull);
            if (counting) {
#pragma unroll
                for (int i = 0; i < 2 * 2 * 2 * 8 * 64 / (4 * TMW_WAVES * 64); ++i) reinterpret_cast<uint4*>(pl)[i * (TMW_WAVES * 64) + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
                (&nc_sh[0][0][0])[threadIdx.x] = 0; (&nc_sh[0][0][0])[TMW_WAVES * 64 + threadIdx.x] = 0;
            }
            __syncthreads();
            const uint32_t s0r = wv ? emid : e0, s1r = wv ? e1 : emid;
            const uint32_t i0 = s0r - base, i1 = s1r - base < tcnt ? s1r - base : tcnt;
            if (counting && i1 > i0) {
                TmState st0, st1; st0.nc = 0; st0.mask = 0; st1.nc = 0; st1.mask = 0;
                tw_range(a, tm, tg, st0, st1, stat, i0, i1, off, tcnt, thr, pkl0, one, lane, K64, kb);
                // the lane's positions: 2 (lane & 31) and the one after it, of tile lane >> 5
                uint32_t* nq = &nc_sh[0][lane >> 5][2 * (lane & 31)];
                if (st0.nc & 0xffffu) atomicAdd(nq, st0.nc & 0xffffu);
                if (st0.nc >> 16) atomicAdd(nq + 128, st0.nc >> 16);
                if (st1.nc & 0xffffu) atomicAdd(nq + 1, st1.nc & 0xffffu);
                if (st1.nc >> 16) atomicAdd(nq + 129, st1.nc >> 16);
            }
            if (threadIdx.x == 0) {
                uint32_t jn2 = NO_JOB;
                if (steal < 8u) {
                    jn2 = job_of(q_q, q_n);
                    if (jn2 == NO_JOB) { if (8u * (q_n >> 6) + q_q >= nblocks) ++steal; jn2 = take(); }
                }
                s_q[2] = jn2;
            }
            const uint32_t c_w0 = w0, c_slab = slab, c_nj = nj, c_tcnt = tcnt; const int32_t c_tstart = tstart; const int c_tid = tid;
            e0 = rl(jw, 0); e1 = rl(jw, 1); w0 = rl(jw, 2); slab = rl(jw, 3); nj = rl(jw, 4); tcnt = rl(jw, 5); win = rl(jw, 6); emid = rl(jw, 7); base = rl(jw, 8); off = rl(jw, 9);
            tstart = (int32_t)rl(jw, 10); tid = (int)rl(jw, 11);
            prefetch();
            __syncthreads();
            const uint32_t nn = rl(s_q[2], 0);
            (void)my_tile;
            if (mine) {                                   // the units of the wave's tile, cell type by cell type
                for (int v = 0; v < 2; ++v) {
                    const int ct = tm.ct_base + v;
                    if (ct >= a.n_ct) break;
                    const uint32_t* pc = pl + v * 1024 + wv * 512;
                    uint32_t dp = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) dp += pc[(lane & 1) * 2048 + k * 64 + (lane >> 1) * 2 + 1] & 0x1fffu;
                    const TwCounters tot{pc, lane, dp - nc_sh[v][wv][lane]};
                    const uint32_t unit = c_w0 + (uint32_t)wv * (uint32_t)a.n_ct + (uint32_t)ct;
                    if (c_nj == 1) {
                        // (the wave writes both cell types' rows, narrow and wide: an arena per cell type and format)
                        if (c_tcnt <= 256u) emit_unit<TwCounters, true>(a, tot, unit, ct, c_tid, c_tstart + 64 * wv, lane, &book, false, refb, 2 * v);
                        else emit_unit<TwCounters, false>(a, tot, unit, ct, c_tid, c_tstart + 64 * wv, lane, &book, false, refb, 2 * v + 1);
                    } else {
                        uint32_t* dst = a.macc + (uint64_t)(c_slab + ((uint32_t)wv * (uint32_t)a.n_ct + (uint32_t)ct) * c_nj) * (NCTR * 64);
                        dst[lane] = tot.NCDUP();
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            dst[(1 + k) * 64 + lane] = tot.DUP(k); dst[(9 + k) * 64 + lane] = tot.BC(k);
                            dst[(17 + k) * 64 + lane] = tot.BQ(k); dst[(25 + k) * 64 + lane] = tot.BCF(k);
                        }
                    }
                }
            }
            if (nxt == NO_JOB) break;
            cur = nxt; nxt = nn;
        }
    }
    lds_fence();
    __syncthreads();
    {
        unsigned long long ev = stat.ev, sg = stat.sg, ne = stat.ne;
        for (int o = 32; o > 0; o >>= 1) { ev += __shfl_down(ev, o); sg += __shfl_down(sg, o); ne += __shfl_down(ne, o); }
        if (lane == 0 && ne) {
            unsigned long long* slot = tg.stat_slots + (size_t)((blockIdx.x * TMW_WAVES + wv) % IX_STAT_SLOTS) * 8;
            atomicAdd(&slot[0], ev); atomicAdd(&slot[1], sg); atomicAdd(&slot[2], ne);
        }
    }
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

// A job longer than TM_JOB_LIMIT entries (a single barcode owning thousands of a tile's entries: its run cannot be cut) does not fit the
// packed planes of k_tm_walk.  One wave per such job, 32-bit planes (quality sum, forward, count, duplicates per symbol and cell type),
// the run logic spelled out: a run start closes the run before it; an entry that is there adds its event where the lane counts it and
// is a duplicate when its symbol was seen in the run already.  Rare by construction; correctness, not speed.
struct WideCounters {
    const uint32_t* pl; int lane; uint32_t ncdup;           // pl: this cell type's [4][8 * 64] planes
    __device__ __forceinline__ uint32_t BQ(int k) const { return pl[k * 64 + lane]; }
    __device__ __forceinline__ uint32_t BCF(int k) const { return pl[512 + k * 64 + lane]; }
    __device__ __forceinline__ uint32_t BC(int k) const { return pl[1024 + k * 64 + lane]; }
    __device__ __forceinline__ uint32_t DUP(int k) const { return pl[1536 + k * 64 + lane]; }
    __device__ __forceinline__ uint32_t NCDUP() const { return ncdup; }
};
// one entry's meta word as k_tm_resolve makes it (the wide walk right after a fused load: nobody has resolved the store yet)
__device__ __forceinline__ uint32_t tm_meta_one(const CountArgs& a, const TmArgs& tm, uint64_t p) {
    const uint32_t s = tm.s0[p], b8 = tm.b[p], cb = s & CB_MASK;
    uint32_t cls = 2;
    bool ok = cb < (uint32_t)a.n_cb;
    if (ok && a.adm) { const uint32_t r = tm.rd[p]; ok = (reinterpret_cast<const uint32_t*>(a.adm)[r >> 5] >> (r & 31u)) & 1u; }
    if (ok && a.drop_pairs && a.read_drop[tm.rd[p]] == 2) ok = !tm_dropped_here(a, tm.rd[p], tm.blk_tile[p >> 3], s);
    if (ok) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
    uint32_t m = cls < 2 ? (cls ? (TMM_CT4 | TMM_CT12) : 0u) | ((s & TM_FWD) ? TMM_FWD : 0u) | ((b8 & 128u) ? TMM_SINGLE : 0u) : TMM_SKIP;
    if (s & TM_RUNSTART) m |= TMM_RS;
    return m;
}
// n_wide: the plan's number of such jobs where it lies on the device (a launch that does not know it yet ends at once when there are
// none), or null; inline_meta: no k_tm_resolve has run for this count
__global__ __launch_bounds__(64) void k_tm_walk_wide(CountArgs a, TmArgs tm, const uint32_t* n_wide, int inline_meta) {
    __shared__ uint32_t pl[2][4][8 * 64];
    __shared__ WaveBook book;
    const int lane = threadIdx.x;
    if (n_wide && rl(*n_wide, 0) == 0u) return;
    book_init(book, lane);
    if (lane == 0) book.src = 2;
    const uint32_t thr = bq_threshold(a);
    const uint16_t* ev16 = reinterpret_cast<const uint16_t*>(tm.store);
    for (uint32_t jx = blockIdx.x; jx < tm.njobs; jx += gridDim.x) {
        const TmJob jb = tm.jobs[jx];
        if (!(jb.nj & TMJ_WIDE) || jb.tile < a.tile_lo || jb.tile >= a.tile_hi) continue;
        const uint32_t nj = jb.nj & ~TMJ_WIDE;
        lds_fence();
        for (int i = lane; i < 2 * 4 * 8 * 64; i += 64) (&pl[0][0][0])[i] = 0;
        lds_fence();
        uint32_t nc[2] = {0u, 0u}, mask = 0, run_ct = 0;
        for (uint32_t p = jb.e0; p < jb.e1; ++p) {
            const uint32_t m = rl(inline_meta ? tm_meta_one(a, tm, p) : tm.meta[p], 0);
            if (m & TMM_RS) { nc[run_ct] += mask ? 1u : 0u; mask = 0; }
            if (m & TMM_SKIP) continue;
            run_ct = (m & TMM_CT4) ? 1u : 0u;
            const uint32_t x = tm.ext[p >> 3];                                        // rows outside the block's extent hold no event (and were never written)
            const uint32_t ev = (uint32_t)lane >= (x & 0xffu) && (uint32_t)lane < (x >> 8) ? ev16[((uint64_t)(p >> 3) * 64 + lane) * 8 + (p & 7u)] : 0u;
            if ((ev & 0x8ffu) >= thr) {
                const uint32_t sym = (ev >> 8) & 7u;
                uint32_t* q = &pl[run_ct][0][sym * 64 + lane];
                q[0] += ev & 0xffu; q[512] += (m & TMM_FWD) ? 1u : 0u; q[1024] += 1u; q[1536] += (mask >> sym) & 1u;
                mask |= 1u << sym;
            }
        }
        nc[run_ct] += mask ? 1u : 0u;
        lds_fence();
        const int2 geom = a.ne_geom[jb.w0];
        const int tid = geom.y & 0xffffff;
        for (int v = 0; v < 2; ++v) {
            const int ct = tm.ct_base + v;
            if (ct >= a.n_ct) break;
            const uint32_t* pc = &pl[v][0][0];
            uint32_t dp = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dp += pc[1024 + k * 64 + lane];
            const WideCounters tot{pc, lane, dp - nc[v]};
            if (nj == 1) emit_unit<WideCounters, false>(a, tot, jb.w0 + ct, ct, tid, geom.x, lane, &book, false);
            else {
                uint32_t* dst = a.macc + (uint64_t)(jb.slab + (uint32_t)ct * nj) * (NCTR * 64);
                dst[lane] = tot.NCDUP();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dst[(1 + k) * 64 + lane] = tot.DUP(k); dst[(9 + k) * 64 + lane] = tot.BC(k);
                    dst[(17 + k) * 64 + lane] = tot.BQ(k); dst[(25 + k) * 64 + lane] = tot.BCF(k);
                }
            }
        }
    }
    book_flush(a, book, lane);
}

// ... and after a load that kept no store: the entry's words from the sort's output, its events from the caller's array
__global__ __launch_bounds__(64) void k_tm_walk_wide_direct(CountArgs a, TmArgs tm, TgArgs tg, const uint32_t* n_wide) {
    __shared__ uint32_t pl[2][4][8 * 64];
    __shared__ WaveBook book;
    const int lane = threadIdx.x;
    if (n_wide && rl(*n_wide, 0) == 0u) return;
    book_init(book, lane);
    if (lane == 0) book.src = 2;
    const uint32_t thr = bq_threshold(a), cbm = (1u << tg.cb_bits) - 1u;
    unsigned long long s_ev = 0, s_sg = 0, s_ne = 0;
    for (uint32_t jx = blockIdx.x; jx < tm.njobs; jx += gridDim.x) {
        const TmJob jb = tm.jobs[jx];
        if (!(jb.nj & TMJ_WIDE) || jb.tile < a.tile_lo || jb.tile >= a.tile_hi) continue;
        const uint32_t nj = jb.nj & ~TMJ_WIDE;
        lds_fence();
        for (int i = lane; i < 2 * 4 * 8 * 64; i += 64) (&pl[0][0][0])[i] = 0;
        lds_fence();
        uint32_t nc[2] = {0u, 0u}, mask = 0, run_ct = 0;
        for (uint32_t p = jb.e0; p < jb.e1 && p - jb.base < jb.cnt; ++p) {
            const uint32_t i = p - jb.base;
            const uint64_t k = tg.key[jb.off + i];
            const uint32_t v = tg.rdv ? tg.rdv[jb.off + i] : (uint32_t)(k >> 32) & (TG_RV_FWD | TG_RV_SEGFIRST), cb = (uint32_t)k & cbm, r = v & TG_RV_READ;
            const bool rs = i == 0 || ((uint32_t)tg.key[jb.off + i - 1] & cbm) != cb;
            const uint32_t geom = (uint32_t)(k >> tg.cb_bits), first = geom & 63u, nev = ((geom >> 6) & 63u) + 1u;
            const uint64_t src = (((k >> (tg.cb_bits + 12)) & tg.src_mask) << tg.src_shift) | (tg.src_shift ? first : 0u);
            uint32_t cls = 2;
            bool ok = cb < (uint32_t)a.n_cb;
            if (ok && a.adm) ok = (reinterpret_cast<const uint32_t*>(a.adm)[r >> 5] >> (r & 31u)) & 1u;
            if (ok) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
            if (rs) { nc[run_ct] += mask ? 1u : 0u; mask = 0; }
            if (cls >= 2) continue;
            s_ev += nev; s_sg += (v & TG_RV_SEGFIRST) ? 1u : 0u; ++s_ne;
            run_ct = cls;
            const uint32_t ev = (uint32_t)lane - first < nev ? tg.events[src + ((uint32_t)lane - first)] : 0u;
            if ((ev & 0x8ffu) >= thr) {
                const uint32_t sym = (ev >> 8) & 7u;
                uint32_t* q = &pl[run_ct][0][sym * 64 + lane];
                q[0] += ev & 0xffu; q[512] += (v & TG_RV_FWD) ? 1u : 0u; q[1024] += 1u; q[1536] += (mask >> sym) & 1u;
                mask |= 1u << sym;
            }
        }
        nc[run_ct] += mask ? 1u : 0u;
        lds_fence();
        const int2 geom = a.ne_geom[jb.w0];
        const int tid = geom.y & 0xffffff;
        for (int v = 0; v < 2; ++v) {
            const int ct = tm.ct_base + v;
            if (ct >= a.n_ct) break;
            const uint32_t* pc = &pl[v][0][0];
            uint32_t dp = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dp += pc[1024 + k * 64 + lane];
            const WideCounters tot{pc, lane, dp - nc[v]};
            if (nj == 1) emit_unit<WideCounters, false>(a, tot, jb.w0 + ct, ct, tid, geom.x, lane, &book, false);
            else {
                uint32_t* dst = a.macc + (uint64_t)(jb.slab + (uint32_t)ct * nj) * (NCTR * 64);
                dst[lane] = tot.NCDUP();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dst[(1 + k) * 64 + lane] = tot.DUP(k); dst[(9 + k) * 64 + lane] = tot.BC(k);
                    dst[(17 + k) * 64 + lane] = tot.BQ(k); dst[(25 + k) * 64 + lane] = tot.BCF(k);
                }
            }
        }
    }
    if (lane == 0 && s_ne) {
        unsigned long long* slot = tg.stat_slots + (size_t)(blockIdx.x % IX_STAT_SLOTS) * 8;
        atomicAdd(&slot[0], s_ev); atomicAdd(&slot[1], s_sg); atomicAdd(&slot[2], s_ne);
    }
    book_flush(a, book, lane);
}

// ... and its wide walk (k_tm_walk_wide_direct over windows: an entry is up to 128 events, the lane takes its position in either tile)
__global__ __launch_bounds__(64) void k_tm_walk_wide_win(CountArgs a, TmArgs tm, TgArgs tg, const uint32_t* n_wide) {
    __shared__ uint32_t pl[2][2][4][8 * 64];               // [tile of the window][cell type][quality sum, forward, count, duplicates][symbol x position]
    __shared__ WaveBook book;
    const int lane = threadIdx.x;
    if (n_wide && rl(*n_wide, 0) == 0u) return;
    book_init(book, lane);
    if (lane == 0) book.src = 2;
    const uint32_t thr = bq_threshold(a), cbm = (1u << tg.cb_bits) - 1u;
    unsigned long long s_ev = 0, s_sg = 0, s_ne = 0;
    for (uint32_t jx = blockIdx.x; jx < tm.njobs; jx += gridDim.x) {
        const TmJob jb = tm.jobs[jx];
        const uint32_t t0 = 2u * jb.tile;
        const bool in_t[2] = {t0 >= a.tile_lo && t0 < a.tile_hi, t0 + 1u >= a.tile_lo && t0 + 1u < a.tile_hi};
        if (!(jb.nj & TMJ_WIDE) || !(in_t[0] || in_t[1])) continue;
        const uint32_t nj = jb.nj & ~TMJ_WIDE;
        lds_fence();
        for (int i = lane; i < 2 * 2 * 4 * 8 * 64; i += 64) (&pl[0][0][0][0])[i] = 0;
        lds_fence();
        uint32_t nc[2][2] = {{0u, 0u}, {0u, 0u}}, mask[2] = {0u, 0u}, run_ct = 0;      // [tile][cell type]; the open run's symbols per tile
        for (uint32_t p = jb.e0; p < jb.e1 && p - jb.base < jb.cnt; ++p) {
            const uint32_t i = p - jb.base;
            const uint64_t k = tg.key[jb.off + i];
            const uint32_t v = (uint32_t)(k >> 32) & (TG_RV_FWD | TG_RV_SEGFIRST), cb = (uint32_t)k & cbm;
            const bool rs = i == 0 || ((uint32_t)tg.key[jb.off + i - 1] & cbm) != cb;
            const uint32_t geom = (uint32_t)(k >> tg.cb_bits), first = geom & 127u, nev = ((geom >> 7) & 127u) + 1u;
            const uint64_t src = (((k >> (tg.cb_bits + 14)) & tg.src_mask) << 7) | first;
            uint32_t cls = 2;
            if (cb < (uint32_t)a.n_cb) { const uint32_t ct = a.celltype_of[cb]; if (ct < (uint32_t)a.n_ct && (ct >> 1) == (uint32_t)(tm.ct_base >> 1)) cls = ct & 1u; }
            if (rs) { for (int h = 0; h < 2; ++h) { nc[h][run_ct] += mask[h] ? 1u : 0u; mask[h] = 0; } }
            if (cls >= 2) continue;
            s_ev += nev; s_sg += (v & TG_RV_SEGFIRST) ? 1u : 0u; ++s_ne;
            run_ct = cls;
            for (int h = 0; h < 2; ++h) {
                const uint32_t q = (uint32_t)(64 * h + lane) - first;
                const uint32_t ev = q < nev ? tg.events[src + q] : 0u;
                if ((ev & 0x8ffu) >= thr) {
                    const uint32_t sym = (ev >> 8) & 7u;
                    uint32_t* w = &pl[h][run_ct][0][sym * 64 + lane];
                    w[0] += ev & 0xffu; w[512] += (v & TG_RV_FWD) ? 1u : 0u; w[1024] += 1u; w[1536] += (mask[h] >> sym) & 1u;
                    mask[h] |= 1u << sym;
                }
            }
        }
        for (int h = 0; h < 2; ++h) nc[h][run_ct] += mask[h] ? 1u : 0u;
        lds_fence();
        for (int h = 0; h < 2; ++h) {
            if (!in_t[h]) continue;
            for (int v = 0; v < 2; ++v) {
                const int ct = tm.ct_base + v;
                if (ct >= a.n_ct) break;
                const uint32_t* pc = &pl[h][v][0][0];
                uint32_t dp = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp += pc[1024 + k * 64 + lane];
                const WideCounters tot{pc, lane, dp - nc[h][v]};
                const uint32_t unit = jb.w0 + (uint32_t)h * (uint32_t)a.n_ct + (uint32_t)ct;
                if (nj == 1) emit_unit<WideCounters, false>(a, tot, unit, ct, jb.tid, jb.tstart + 64 * h, lane, &book, false, -1, v);
                else {
                    uint32_t* dst = a.macc + (uint64_t)(jb.slab + ((uint32_t)h * (uint32_t)a.n_ct + (uint32_t)ct) * nj) * (NCTR * 64);
                    dst[lane] = tot.NCDUP();
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        dst[(1 + k) * 64 + lane] = tot.DUP(k); dst[(9 + k) * 64 + lane] = tot.BC(k);
                        dst[(17 + k) * 64 + lane] = tot.BQ(k); dst[(25 + k) * 64 + lane] = tot.BCF(k);
                    }
                }
            }
        }
    }
    if (lane == 0 && s_ne) {
        unsigned long long* slot = tg.stat_slots + (size_t)(blockIdx.x % IX_STAT_SLOTS) * 8;
        atomicAdd(&slot[0], s_ev); atomicAdd(&slot[1], s_sg); atomicAdd(&slot[2], s_ne);
    }
    book_flush(a, book, lane);
}

// Everything a count needs before its first kernel: the plan's unit tables where the call stage and the exports read them, row
// buffers, zeroed counters, the kernels' arguments; k_read_stats is queued (the admitted reads, and the bit per read the entries'
// admission is looked up in when some stored read can fail THIS count's read filters).
struct CountLaunch { CountArgs a; TmArgs tm; unsigned grid_walk, grid_fin; int n_pass; };
// a handful of fills and copies in one launch (count_prepare): word granularity, every buffer a device allocation of its own
struct PrepOp { void* dst; const void* src; uint64_t words; };      // src null: zeros
struct PrepArgs { PrepOp op[10]; int n; };
__global__ __launch_bounds__(256) void k_prep_ops(PrepArgs a) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (uint64_t)gridDim.x * blockDim.x;
    for (int k = 0; k < a.n; ++k) {
        uint32_t* d = reinterpret_cast<uint32_t*>(a.op[k].dst);
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(a.op[k].src);
        const uint64_t nq = a.op[k].words / 4;
        for (uint64_t i = g; i < nq; i += gs) reinterpret_cast<uint4*>(d)[i] = sp ? reinterpret_cast<const uint4*>(sp)[i] : make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t i = nq * 4 + g; i < a.op[k].words; i += gs) d[i] = sp ? sp[i] : 0u;
    }
}
static int count_prepare(lsg_ctx* c, const lsg_count_params* p, CountLaunch& L) {
    hipStream_t st = c->stream;
    const uint32_t n_ne = c->tm_n_ne;
    if (c->d_scalars.reserve(SC_COUNT * 8) || c->d_ne_units.reserve(((size_t)n_ne + 2) * 4) ||
        c->d_ne_mask.reserve(((size_t)n_ne + 2) * 8) || c->d_ne_rowbase.reserve(((size_t)n_ne + 2) * 4) || c->ws[WS_NE_NSLOT].reserve(((size_t)n_ne + 2) * 4) ||
        c->ws[WS_NE_ACC].reserve(((size_t)n_ne + 2) * 4) || c->ws[WS_NE_GEOM].reserve(((size_t)n_ne + 2) * 8) || c->ws[WS_MULTI_LIST].reserve(((size_t)c->tm_n_multi + 2) * 4) ||
        c->ws[WS_MACC].reserve(((size_t)c->tm_n_slabs + 1) * NCTR * 64 * 4) || c->d_ix_stat.reserve(IX_STAT_SLOTS * 64) ||
        c->d_read_adm.reserve(((size_t)c->rd.n_reads / 64 + 2) * 8))
        return -1;
    // 16 workgroups = 32 waves per CU = the 8 waves per SIMD the hardware holds (8 KB of LDS each): the walk waits on its own dependency
    // chains (a scalar decision per entry), so every resident wave counts — 14 per CU: 5.7 ms, 16: 5.4 (round 3)
    L.grid_walk = (unsigned)(c->n_cus * tune_int("LSG_GRID_TM", 16));
    L.grid_fin = (unsigned)(c->n_cus * 8);
    L.n_pass = (c->n_ct + 1) / 2;
    {   // row buffers: bound + one open arena per emitting wave and format
        uint64_t want_rows = (uint64_t)n_ne / (uint64_t)c->n_ct * TILE_W;
        if (p->min_dp > 0) {
            const uint64_t by_depth = (uint64_t)c->rd.n_events / (uint64_t)p->min_dp + 64;
            if (by_depth < want_rows) want_rows = by_depth;
        }
        const uint64_t emitters = (uint64_t)L.grid_walk * TMW_WAVES * 2 + L.grid_fin + (uint64_t)c->n_cus * 4 + (c->wsh ? (uint64_t)c->n_cus * 16 * TMW_WAVES * 4 : 0u);      // (the windows' count: four arenas a wave)
        uint64_t arena = want_rows / (emitters * 8) / ARENA * ARENA;
        c->arena = (uint32_t)(arena < (uint64_t)ARENA ? (uint64_t)ARENA : (arena > 8ull * ARENA ? 8ull * ARENA : arena));
        want_rows += emitters * c->arena + 64;
        want_rows = (want_rows + 63) / 64 * 64 + 64;        // whole 64-row blocks (lsg::row_word), one spare: a unit's descriptor spans two
        if (want_rows > c->row_cap) c->row_cap = want_rows;
        for (int i = 0; i < c->n_ct; ++i)
            if (c->d_rows[i].reserve((size_t)c->row_cap * ROW_STORED_WORDS * 4)) return -1;
    }
    c->n_ne = n_ne; c->n_multi = c->tm_n_multi;
    L.a = CountArgs{};
    fill_args(c, p, L.a);
    TmArgs& tm = L.tm;
    tm = TmArgs{};
    tm.store = c->tm[TM_STORE].as<uint4>(); tm.s0 = c->tm[TM_S0].as<uint32_t>(); tm.b = c->tm[TM_B].as<uint8_t>(); tm.rd = c->tm[TM_RD].as<uint32_t>();
    tm.meta = c->tm[TM_META].as<uint32_t>(); tm.blk_tile = c->tm[TM_BLK_TILE].as<uint32_t>(); tm.jobs = c->tm[TM_JOBS].as<TmJob>(); tm.np = c->tm_np; tm.nblk = c->tm_nblk;
    tm.njobs = c->tm_njobs; tm.nchunks = c->tm_nchunks; tm.chunk_start = c->tm[TM_CHUNKS].as<uint32_t>(); tm.ext = c->tm[TM_EXT].as<uint16_t>();
    LSG_HIP(hipEventRecord(c->ev[0], st));
    {   // zeroed counters, and the static unit tables in the places the call stage and the exports read: ONE launch (nine fills and copies of
        // a few megabytes were nine commands, ~10 us each with the gaps between them, in front of every count)
        PrepArgs pa{}; int k = 0;
        auto op = [&](void* dst, const void* src, size_t bytes) { if (bytes) { pa.op[k].dst = dst; pa.op[k].src = src; pa.op[k].words = (bytes + 3) / 4; ++k; } };
        op(c->d_scalars.p, nullptr, SC_COUNT * 8);
        op(c->d_ix_stat.p, nullptr, IX_STAT_SLOTS * 64);
        if (n_ne) {
            op(c->d_ne_units.p, c->tm[TM_NE_UNITS].p, (size_t)n_ne * 4);
            op(c->ws[WS_NE_GEOM].p, c->tm[TM_NE_GEOM].p, (size_t)n_ne * 8);
            op(c->ws[WS_NE_NSLOT].p, c->tm[TM_NE_NSLOT].p, ((size_t)n_ne + 1) * 4);
            op(c->ws[WS_NE_ACC].p, c->tm[TM_NE_ACC].p, ((size_t)n_ne + 1) * 4);
            if (c->tm_n_multi) op(c->ws[WS_MULTI_LIST].p, c->tm[TM_MULTI].p, (size_t)c->tm_n_multi * 4);
            op(c->d_ne_mask.p, nullptr, ((size_t)n_ne + 1) * 8);
            op(c->d_ne_rowbase.p, nullptr, ((size_t)n_ne + 1) * 4);
        }
        pa.n = k;
        hipLaunchKernelGGL(k_prep_ops, dim3((unsigned)(c->n_cus * 4)), dim3(256), 0, st, pa);
    }
    const int64_t R = c->rd.n_reads;
    if (R > 0) { unsigned g = (unsigned)((R + 255) / 256); if (g > (unsigned)(c->n_cus * 8)) g = (unsigned)(c->n_cus * 8); hipLaunchKernelGGL(k_read_stats, dim3(g), dim3(256), 0, st, L.a); }
    return 0;
}
// the passes over the resident store: resolve + walk per pair of cell types (from pass `first_pass` on), the wide walk where the plan has
// jobs for it
static int count_passes(lsg_ctx* c, CountLaunch& L, int first_pass, bool wide_of_pass0) {
    hipStream_t st = c->stream;
    const unsigned grid_wide = c->tm_n_wide ? (c->tm_n_wide < (unsigned)(c->n_cus * 4) ? c->tm_n_wide : (unsigned)(c->n_cus * 4)) : 0u;
    for (int pass = wide_of_pass0 ? 0 : first_pass; pass < L.n_pass && c->tm_nblk; ++pass) {
        const bool wide_only = pass < first_pass;             // (the fused load counted this pass's jobs but the wide ones)
        if (wide_only && !grid_wide) continue;
        L.tm.ct_base = 2 * pass;
        if (pass) LSG_HIP(hipMemsetAsync(L.a.scalars + SC_QWALK, 0, 8, st));          // the walk's chunk queue starts over
        if (L.a.drop_pairs) hipLaunchKernelGGL(k_tm_resolve<true>, dim3((c->tm_nblk + 255) / 256), dim3(256), 0, st, L.a, L.tm, wide_only ? (unsigned long long*)nullptr : c->d_ix_stat.as<unsigned long long>());
        else hipLaunchKernelGGL(k_tm_resolve<false>, dim3((c->tm_nblk + 255) / 256), dim3(256), 0, st, L.a, L.tm, wide_only ? (unsigned long long*)nullptr : c->d_ix_stat.as<unsigned long long>());
        if (pass == 0) { LSG_HIP(hipEventRecord(c->ev[1], st)); LSG_HIP(hipEventRecord(c->ev[3], st)); }
        if (c->tm_njobs && !wide_only) hipLaunchKernelGGL(k_tm_walk, dim3(L.grid_walk), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm);
        if (pass == 0) LSG_HIP(hipEventRecord(c->ev[4], st));
        if (grid_wide) hipLaunchKernelGGL(k_tm_walk_wide, dim3(grid_wide), dim3(64), 0, st, L.a, L.tm, (const uint32_t*)nullptr, 0);
    }
    return 0;
}
// the count's tail: statistics, multi-job tiles, the counters' way to the host
static int count_finish(lsg_ctx* c, const lsg_count_params* p, CountLaunch& L) {
    hipStream_t st = c->stream;
    if (c->tm_nblk) hipLaunchKernelGGL(k_resolve_stats, dim3(1), dim3(256), 0, st, L.a, c->d_ix_stat.as<unsigned long long>());
    if (c->tm_n_multi) hipLaunchKernelGGL(k_finalize_multi, dim3(c->tm_n_multi < L.grid_fin ? c->tm_n_multi : L.grid_fin), dim3(FIN_THREADS), 0, st, L.a);
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
    c->stats.n_units = c->n_ne;
    c->stats.n_deep_units = c->tm_n_multi;
    c->stats.n_entries = (int64_t)sc[SC_NENT];
    float ms = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->stats.ms_bin = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[1], c->ev[5])); c->stats.ms_deep = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); c->stats.ms_walk = ms;
    c->stats.ms_wave = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[5])); c->stats.ms_total = ms;
    for (int i = 0; i < 4; ++i) { c->stats.rows_by_kernel[i] = (int64_t)sc[SC_ROWS_SRC + i]; c->stats.events_by_kernel[i] = 0; }
    c->stats.events_by_kernel[1] = (int64_t)sc[SC_EVENTS];           // the walk reads every event
    c->stats.n_events_wave = 0; c->stats.n_events_deep = (int64_t)sc[SC_EVENTS];
    c->stats.n_rows_deep = (int64_t)sc[SC_ROWS_DEEP];
    c->stats.n_rows_wave = -c->stats.n_rows_deep;
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->stats.n_rows_wave += c->n_rows[i];
    c->last_params = *p;
    c->counted = true;
    c->called = false;
    ++c->count_serial;
    return 0;
}

// The fused pass of a load (store.hip build_store when lsg_set_count_at_load is on): the plan is made (its chunk and wide-job counters
// still on the device), the sort's output lies in src; queues everything a count queues around k_tm_gather_count and ends with the
// count's one synchronisation.  At most two cell types (one pass), no depth-cap drops (the caller has checked the bound).
int run_gather_count(lsg_ctx* c, const lsg_count_params* p, const GatherCountSrc& src, bool direct) {
    hipStream_t st = c->stream;
    c->has_drops = false; c->n_depth_dropped = 0;
    CountLaunch L;
    if (int rc = count_prepare(c, p, L)) return rc;
    TgArgs tg{};
    tg.events = src.events; tg.n_events = src.n_events; tg.key = src.key; tg.rdv = src.rdv; tg.cb_bits = src.cb_bits;
    tg.src_mask = (1ull << ((src.rdv ? 52 : 50) - src.cb_bits)) - 1ull; tg.src_shift = src.src_shift;
    if (!src.rdv && (!direct || L.a.adm)) return 1;      // (keys alone carry no read index: a count that looks reads up is made from a load that kept them; build_store loads again)
    tg.tile_off = c->d_tile_off.as<uint32_t>(); tg.blk_off = c->tm[TM_BLK_OFF].as<uint32_t>();
    tg.s0 = c->tm[TM_S0].as<uint32_t>(); tg.b8 = c->tm[TM_B].as<uint8_t>(); tg.rd = c->tm[TM_RD].as<uint32_t>();
    tg.store = c->tm[TM_STORE].as<uint4>(); tg.ext = c->tm[TM_EXT].as<uint16_t>();
    tg.nchunks = c->d_plan_misc + 1; tg.stat_slots = c->d_ix_stat.as<unsigned long long>();
    L.tm.ct_base = 0;
    const bool dbg = getenv("LSG_DEBUG_SYNC") != nullptr;
    auto stage = [&](const char* what) { if (dbg) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[lsg] fused load: %s: %s\n", what, hipGetErrorString(e)); fflush(stderr); } };
    stage("count prepared");
    LSG_HIP(hipEventRecord(c->ev[1], st)); LSG_HIP(hipEventRecord(c->ev[3], st));
    if (direct) {
        // no store (lsg_set_store_policy): 80 registers per lane: 6 waves per SIMD = 12 workgroups per CU (8: 9.3 ms, 10: 8.5, 12: 8.2)
        if (c->d_xcd_queues.reserve(8 * 128)) return -1;
        LSG_HIP(hipMemsetAsync(c->d_xcd_queues.p, 0, 8 * 128, st));
        tg.queues = c->d_xcd_queues.as<unsigned long long>();
        if (src.wsh) {
            // entries binned by 128-position windows (store.hip build_store): one 256-byte block per entry; 17 KB of LDS a workgroup: 9 per CU
            if (tg.rdv || L.a.adm || p->min_bq < 1 || p->min_bq > 255 || tg.src_shift != 7) return 1;      // (build_store makes such a load by tiles)
            tg.src_mask = (1ull << (48 - src.cb_bits)) - 1ull;      // (seven bits each for the first position and the events - 1)
            c->line_loads = true;
            const unsigned gridw = (unsigned)(c->n_cus * tune_int("LSG_GRID_TW", 9));
            hipLaunchKernelGGL(k_tm_count_win, dim3(gridw), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
            LSG_HIP(hipEventRecord(c->ev[4], st));
            LSG_HIP(hipEventRecord(c->evb[4], st));
            stage("k_tm_count_win");
            hipLaunchKernelGGL(k_tm_walk_wide_win, dim3((unsigned)(c->n_cus * 2)), dim3(64), 0, st, L.a, L.tm, tg, (const uint32_t*)c->d_plan_misc);
            stage("wide walk");
            return count_finish(c, p, L);
        }
        const unsigned grid = (unsigned)(c->n_cus * tune_int("LSG_GRID_TD", 12));
        // tile-phased events: an entry is fetched as its one 128-byte line (tm_add<.., true>: the quality compare reads the event's low byte)
        const bool al = tg.src_shift == 6 && p->min_bq >= 1 && p->min_bq <= 255 && !getenv("LSG_NO_LINE_LOADS");
        c->line_loads = al;
        if (tg.rdv && al) hipLaunchKernelGGL((k_tm_count_direct<false, true>), dim3(grid), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
        else if (tg.rdv) hipLaunchKernelGGL((k_tm_count_direct<false, false>), dim3(grid), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
        else if (al) hipLaunchKernelGGL((k_tm_count_direct<true, true>), dim3(grid), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
        else hipLaunchKernelGGL((k_tm_count_direct<true, false>), dim3(grid), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
        LSG_HIP(hipEventRecord(c->ev[4], st));
        LSG_HIP(hipEventRecord(c->evb[4], st));
        stage("k_tm_count_direct");
        hipLaunchKernelGGL(k_tm_walk_wide_direct, dim3((unsigned)(c->n_cus * 2)), dim3(64), 0, st, L.a, L.tm, tg, (const uint32_t*)c->d_plan_misc);
        stage("wide walk");
        return count_finish(c, p, L);
    }
    // 17 KB of LDS per workgroup (planes + the two waves' transposition tiles): 9 workgroups per CU
    const unsigned grid = (unsigned)(c->n_cus * tune_int("LSG_GRID_TG", 9));
    hipLaunchKernelGGL(k_tm_gather_count, dim3(grid), dim3(TMW_WAVES * 64), 0, st, L.a, L.tm, tg);
    LSG_HIP(hipEventRecord(c->ev[4], st));
    LSG_HIP(hipEventRecord(c->evb[4], st));
    stage("k_tm_gather_count");
    // jobs too long for the packed planes were gathered, not counted: the wide walk takes them from the store (it ends at once when the plan has none)
    hipLaunchKernelGGL(k_tm_walk_wide, dim3((unsigned)(c->n_cus * 2)), dim3(64), 0, st, L.a, L.tm, (const uint32_t*)c->d_plan_misc, 1);
    stage("wide walk");
    return count_finish(c, p, L);
}

int run_count(lsg_ctx* c, const lsg_count_params* p) {
    if (c->n_contigs <= 0) { set_error("lsg_pileup_count: no contigs set"); return -2; }
    if (c->n_ct <= 0) { set_error("lsg_pileup_count: no barcodes set"); return -2; }
    if (c->store_skipped) { set_error("lsg_pileup_count: the load kept no store (lsg_set_store_policy): only the count it made is there - load the reads again to count with other parameters, tables or regions"); return -2; }
    if (!c->tm_valid) { set_error("lsg_pileup_count: no reads loaded"); return -2; }
    for (int t = 0; t < c->n_contigs; ++t)
        if (!c->ref_ptr[t]) { set_error("lsg_pileup_count: reference of contig %d not loaded", t); return -2; }
    if (int rc = ensure_plan(c)) return rc;
    if (depth_cap_drops(c, p)) return -1;           // free unless some cell type's pileup buffer can reach max_depth (cached bound)
    CountLaunch L;
    if (int rc = count_prepare(c, p, L)) return rc;
    hipStream_t st = c->stream;
    if (int rc = count_passes(c, L, 0, false)) return rc;
    if (!c->tm_nblk) { LSG_HIP(hipEventRecord(c->ev[1], st)); LSG_HIP(hipEventRecord(c->ev[3], st)); LSG_HIP(hipEventRecord(c->ev[4], st)); }
    c->counted_at_load = false;
    return count_finish(c, p, L);
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

// The count rows of one cell type as flat device arrays (keys (tid << 32) | pos0, reference bases, LSG_ROW_WORDS counters per row), in
// genomic order: what lsg_fetch_counts copies to the host and tables.hip prints.
int run_export_rows(lsg_ctx* c, int ct, DevBuf& dk, DevBuf& dr, DevBuf& dc) {
    int64_t n = c->n_rows[ct];
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
    if (dk.reserve((size_t)n * 8) || dr.reserve((size_t)n) || dc.reserve((size_t)n * LSG_ROW_WORDS * 4)) return -1;
    uint64_t threads = (uint64_t)n_ne * 64;
    hipLaunchKernelGGL(k_export_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, ct, off,
                       dk.as<int64_t>(), dr.as<uint8_t>(), dc.as<uint32_t>());
    LSG_HIP(hipGetLastError());
    return 0;
}

int run_fetch_counts(lsg_ctx* c, int ct, int64_t* keys, uint8_t* ref, uint32_t* counts, int64_t capacity) {
    if (!c->counted) { set_error("lsg_fetch_counts: call lsg_pileup_count first"); return -2; }
    if (ct < 0 || ct >= c->n_ct) { set_error("lsg_fetch_counts: bad cell type %d", ct); return -2; }
    int64_t n = c->n_rows[ct];
    if (capacity < n) { set_error("lsg_fetch_counts: capacity %lld < %lld rows", (long long)capacity, (long long)n); return -2; }
    if (n == 0) return 0;
    hipStream_t st = c->stream;
    DevBuf &dk = c->ws[WS_EXPORT_K], &dr = c->ws[WS_EXPORT_R], &dc = c->ws[WS_EXPORT_C];
    if (int rc = run_export_rows(c, ct, dk, dr, dc)) return rc;
    LSG_HIP(hipMemcpyAsync(keys, dk.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(ref, dr.p, (size_t)n, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(counts, dc.p, (size_t)n * LSG_ROW_WORDS * 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    return 0;
}

} // namespace lsg
