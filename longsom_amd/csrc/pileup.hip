// Per-cell-type pileup base counting on CDNA4 (gfx950).
//
// Replaces, for every covered column at once:
//   split_bam's read routing            workflow/scripts/PreProcessing/SplitBamCellTypes.py:65-124
//   run_interval (pileup + counting)    workflow/scripts/SNVCalling/BaseCellCounter.py:182-320
//
// Work decomposition:
//   unit  = (64-position tile of one contig, cell type); one lane per reference position.
//   entry = one read segment overlapping a unit; built by a two-pass counting sort (k_count_units,
//           scan, k_scatter_entries).
//   A unit's entries are grouped by cell barcode in LDS (open-addressing hash + scan), then walked
//   barcode-run by barcode-run: every entry is one coalesced 128-byte load of uint16 events
//   (lane = position), counters live in registers / lane-private LDS words, and the number of
//   distinct cells (NC, CC[sym]) is the number of barcode runs with the symbol present.  No global
//   atomics on the event path, integer arithmetic only (HBM-bound; no MFMA).
//   Units with <= CAPW entries are processed by one wavefront each (no block barriers); deeper
//   units (highly expressed genes, chrM) by a whole workgroup in barcode-range passes.
#include "lsg_ctx.h"
#include <hipcub/hipcub.hpp>

namespace lsg {

constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;
constexpr uint32_t CB_MASK = 0x00FFFFFFu;
constexpr int CAPW = 256;            // max entries of a wave-processed unit
constexpr int HW = 512;              // hash slots of the wave kernel (2 x CAPW)
constexpr int CAPB = 2048;           // staged entries per pass of the deep (workgroup) kernel
constexpr int HB = 4096;             // hash slots of the deep kernel
constexpr int NBUCKET = 1024;        // coarse barcode buckets of the deep kernel
constexpr int DEEP_THREADS = 256;

// device scalars (uint64 each)
enum { SC_QHEAD = 0, SC_QHEAD_DEEP = 1, SC_NDEEP = 3, SC_ROWS = 4, SC_COLS = 8, SC_OVERFLOW = 9,
       SC_READS = 10, SC_SEGS = 11, SC_EVENTS = 12, SC_COUNT = 16 };

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
    // params
    int32_t min_bq, min_mq, min_dp, min_cc, ignore_orphans;
    uint32_t flag_exclude;
    // workspace
    uint32_t* read_key; uint32_t* unit_cnt; uint32_t* unit_off; uint32_t* unit_cursor;
    uint2* entries;
    uint32_t* ne_units; const uint32_t* n_ne; uint64_t* ne_mask; uint32_t* ne_rowbase;
    uint32_t* deep_list;
    unsigned long long* scalars;
    uint32_t* rows[LSG_MAX_CELLTYPES];
    uint64_t row_cap;
};

// ------------------------------------------------------------------------------------------------
// Read admission = the union of the reference's filters on the count path:
//   pysam pileup flag_filter (UNMAP|SECONDARY|QCFAIL|DUP) and min_mapping_quality, ignore_orphans
//   (BaseCellCounter.py:191), is_supplementary (:249), CB tag present (:240-243), CB in barcodes.tsv
//   with a cell type (SplitBamCellTypes.py:83-90), MAPQ >= min_MQ (:110-113).
// key = cb | reverse<<24 | celltype<<28.
__global__ void k_read_key(CountArgs a) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    uint32_t key = KEY_INVALID;
    uint32_t flag = a.read_flag[r];
    int32_t cb = a.read_cb[r];
    int32_t tid = a.read_tid[r];
    bool ok = (flag & a.flag_exclude) == 0 && (int)a.read_mapq[r] >= a.min_mq && cb >= 0 && cb < a.n_cb &&
              tid >= 0 && tid < a.n_contigs;
    if (ok && a.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) ok = false;
    if (ok) {
        uint32_t ct = a.celltype_of[cb];
        if (ct < (uint32_t)a.n_ct) key = (uint32_t)cb | (((flag >> 4) & 1u) << 24) | (ct << 28);
    }
    a.read_key[r] = key;
    unsigned long long m = __ballot(key != KEY_INVALID);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.scalars[SC_READS], (unsigned long long)__popcll(m));
}

template <bool SCATTER>
__global__ void k_bin_segments(CountArgs a) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = s < a.n_segs;
    uint32_t key = KEY_INVALID;
    int32_t tid = 0, st = 0, ln = 0;
    if (live) {
        uint32_t r = a.seg_read[s];
        key = a.read_key[r];
        tid = a.read_tid[r];
        st = a.seg_start[s];
        ln = a.seg_len[s];
        if (key != KEY_INVALID) {
            int64_t clen = a.contig_len[tid];
            if (st < 0 || ln <= 0 || (int64_t)st + ln > clen) key = KEY_INVALID;   // malformed: never counted
        }
    }
    bool ok = key != KEY_INVALID;
    if (!SCATTER) {
        unsigned long long m = __ballot(ok);
        unsigned long long evs = ok ? (unsigned long long)ln : 0ull;
        for (int o = 32; o > 0; o >>= 1) evs += __shfl_down(evs, o);
        if ((threadIdx.x & 63) == 0 && m) {
            atomicAdd(&a.scalars[SC_SEGS], (unsigned long long)__popcll(m));
            atomicAdd(&a.scalars[SC_EVENTS], evs);
        }
    }
    if (!ok) return;
    uint32_t ct = key >> 28;
    uint32_t t0 = a.tile_base[tid] + ((uint32_t)st >> 6);
    uint32_t t1 = a.tile_base[tid] + ((uint32_t)(st + ln - 1) >> 6);
    for (uint32_t t = t0; t <= t1; ++t) {
        uint32_t u = t * (uint32_t)a.n_ct + ct;
        if (SCATTER) {
            uint32_t slot = atomicAdd(&a.unit_cursor[u], 1u);
            a.entries[slot] = make_uint2(key, (uint32_t)s);
        } else {
            atomicAdd(&a.unit_cnt[u], 1u);
        }
    }
}

__global__ void k_deep_list(CountArgs a) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= *a.n_ne) return;
    uint32_t u = a.ne_units[w];
    if (a.unit_cnt[u] > (uint32_t)CAPW) {
        unsigned long long i = atomicAdd(&a.scalars[SC_NDEEP], 1ull);
        a.deep_list[i] = w;
    }
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
__device__ __forceinline__ uint32_t spread4(uint32_t m) { return (m * 0x00204081u) & 0x01010101u; }
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// Per-lane (= per reference position) accumulators of one unit.
struct Acc {
    uint32_t bc[8], bq[8], bcf[8], cc[8], nc;
    uint32_t cclo, cchi, mask, npk, nruns, prev_cb;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < 8; ++s) bc[s] = bq[s] = bcf[s] = cc[s] = 0;
        nc = cclo = cchi = mask = npk = nruns = 0; prev_cb = KEY_INVALID;
    }
    __device__ __forceinline__ void flush_pk(uint32_t* pk, int lane) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint32_t v = pk[s * 64 + lane];
            pk[s * 64 + lane] = 0;
            bcf[s] += v & 0xffu; bc[s] += (v >> 8) & 0xffu; bq[s] += v >> 16;
        }
        npk = 0;
    }
    __device__ __forceinline__ void flush_cc() {
#pragma unroll
        for (int s = 0; s < 4; ++s) { cc[s] += (cclo >> (8 * s)) & 0xffu; cc[4 + s] += (cchi >> (8 * s)) & 0xffu; }
        cclo = cchi = 0; nruns = 0;
    }
    // A barcode run ends: each symbol seen in the run is one more distinct cell
    // (len(set(CELL_COUNTS[x])), len(set(CELLS)): BaseCellCounter.py:283,292).
    __device__ __forceinline__ void close_run() {
        cclo += spread4(mask & 15u); cchi += spread4(mask >> 4); nc += mask != 0; mask = 0;
        if (++nruns == 255) flush_cc();
    }
    // One pileup entry at this lane's position (BaseCellCounter.py:258-279).
    __device__ __forceinline__ void add(uint32_t ev, bool in_range, uint32_t fwd, int min_bq, uint32_t* pk, int lane) {
        uint32_t q = ev & 0xffu, sym = ev >> 8;
        if (in_range && sym < 8 && (int)q >= min_bq) {
            atomicAdd(&pk[sym * 64 + lane], (q << 16) | 0x100u | fwd);   // ds_add_u32, lane-private word
            mask |= 1u << sym;
        }
        if (++npk == 255) flush_pk(pk, lane);
    }
    __device__ __forceinline__ void finish(uint32_t* pk, int lane) { close_run(); flush_cc(); flush_pk(pk, lane); }
};

struct Rec { uint32_t key, ev_lo, meta; };   // meta = ev_hi(8) | lane_lo(6)<<8 | (cnt-1)(6)<<16

__device__ __forceinline__ Rec make_rec(const CountArgs& a, uint2 e, int32_t tstart) {
    uint32_t s = e.y;
    int32_t st = a.seg_start[s], ln = a.seg_len[s];
    int64_t off = a.seg_ev_off[s];
    int32_t lo = st > tstart ? st : tstart;
    int32_t hi = st + ln < tstart + TILE_W ? st + ln : tstart + TILE_W;
    int64_t ev_first = off + (lo - st);
    Rec r;
    r.key = e.x;
    r.ev_lo = (uint32_t)ev_first;
    r.meta = (uint32_t)((ev_first >> 32) & 0xff) | ((uint32_t)(lo - tstart) << 8) | ((uint32_t)(hi - lo - 1) << 16);
    return r;
}

// Walk entries order[j0..j1) (grouped by barcode) with all 64 lanes = 64 positions.
__device__ __forceinline__ void walk(const CountArgs& a, Acc& acc, const uint16_t* order, const uint32_t* rkey,
                                     const uint32_t* rev, const uint32_t* rmeta, int j0, int j1, uint32_t* pk, int lane) {
    const uint16_t* __restrict__ events = a.events;
    for (int jb = j0; jb < j1; jb += 64) {
        int nb = j1 - jb < 64 ? j1 - jb : 64;
        uint32_t k = 0, e = 0, m = 0;
        if (lane < nb) { uint32_t idx = order[jb + lane]; k = rkey[idx]; e = rev[idx]; m = rmeta[idx]; }
        int l = 0;
        for (; l + 4 <= nb; l += 4) {
            uint32_t ks[4], evv[4]; bool inr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ks[u] = rl(k, l + u);
                uint32_t es = rl(e, l + u), ms = rl(m, l + u);
                uint32_t rel = (uint32_t)lane - ((ms >> 8) & 63u);
                inr[u] = rel <= ((ms >> 16) & 63u);
                uint64_t addr = (((uint64_t)(ms & 0xffu)) << 32 | es) + rel;
                evv[u] = inr[u] ? (uint32_t)events[addr] : 0xffffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t cb = ks[u] & CB_MASK;
                if (cb != acc.prev_cb) { acc.close_run(); acc.prev_cb = cb; }
                acc.add(evv[u], inr[u], ((ks[u] >> 24) & 1u) ^ 1u, a.min_bq, pk, lane);
            }
        }
        for (; l < nb; ++l) {
            uint32_t ks = rl(k, l), es = rl(e, l), ms = rl(m, l);
            uint32_t rel = (uint32_t)lane - ((ms >> 8) & 63u);
            bool inr = rel <= ((ms >> 16) & 63u);
            uint64_t addr = (((uint64_t)(ms & 0xffu)) << 32 | es) + rel;
            uint32_t evv = inr ? (uint32_t)events[addr] : 0xffffu;
            uint32_t cb = ks & CB_MASK;
            if (cb != acc.prev_cb) { acc.close_run(); acc.prev_cb = cb; }
            acc.add(evv, inr, ((ks >> 24) & 1u) ^ 1u, a.min_bq, pk, lane);
        }
    }
}

// Group n staged records by barcode: order[] lists record indices with equal barcodes adjacent.
// T threads cooperate (T = 64: one wave, fences only; T = DEEP_THREADS: __syncthreads).
template <bool BLOCK, int H>
__device__ __forceinline__ void group_by_cb(int n, const uint32_t* rkey, uint32_t* tkey, uint32_t* tcnt, uint16_t* order,
                                            int t, int T, uint32_t* wave_tot) {
    for (int i = t; i < H; i += T) { tkey[i] = KEY_INVALID; tcnt[i] = 0; }
    group_sync<BLOCK>();
    constexpr int RMAX = BLOCK ? (CAPB + DEEP_THREADS - 1) / DEEP_THREADS : (CAPW + 63) / 64;
    uint32_t hs[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        int i = t + r * T;
        hs[r] = 0;
        if (i < n) {
            uint32_t cb = rkey[i] & CB_MASK;
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
    constexpr int PER = BLOCK ? H / DEEP_THREADS : H / 64;
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
        int i = t + r * T;
        if (i < n) order[tcnt[hs[r] & 0xffffu] + (hs[r] >> 16)] = (uint16_t)i;
    }
    group_sync<BLOCK>();
}

__device__ __forceinline__ void unit_geometry(const CountArgs& a, uint32_t u, int& ct, int& tid, int32_t& tstart) {
    uint32_t tile = u / (uint32_t)a.n_ct;
    ct = (int)(u - tile * (uint32_t)a.n_ct);
    int lo = 0, hi = a.n_contigs;                 // largest tid with tile_base[tid] <= tile
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.tile_base[mid] <= tile) lo = mid; else hi = mid; }
    tid = lo;
    tstart = (int32_t)((tile - a.tile_base[tid]) << 6);
}

// Gates + row emission for one unit by one wave.  Gates: BaseCellCounter.py:211 (ref != N), :282
// (count >= MIN_COV), :294 (NC >= MIN_CC); position 0 of a contig is never visited (:86).
__device__ __forceinline__ void emit_unit(const CountArgs& a, const Acc& acc, uint32_t w, int ct, int tid, int32_t tstart, int lane) {
    uint32_t dp = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) dp += acc.bc[s];
    int64_t pos = (int64_t)tstart + lane;
    bool valid = pos >= 1 && pos < a.contig_len[tid];
    uint8_t refb = 'N';
    const uint8_t* rp = a.ref_ptr[tid];
    if (valid) refb = rp ? rp[pos] : (uint8_t)'?';
    unsigned long long colm = __ballot(valid && dp > 0);
    bool emit = valid && dp > 0 && (int)dp >= a.min_dp && (int)acc.nc >= a.min_cc && refb != 'N';
    unsigned long long em = __ballot(emit);
    uint32_t base = 0;
    if (lane == 0) {
        if (colm) atomicAdd(&a.scalars[SC_COLS], (unsigned long long)__popcll(colm));
        if (em) base = (uint32_t)atomicAdd(&a.scalars[SC_ROWS + ct], (unsigned long long)__popcll(em));
        a.ne_mask[w] = em;
        a.ne_rowbase[w] = base;
    }
    base = rl(base, 0);
    if (!em) return;
    uint64_t row = (uint64_t)base + __popcll(em & ((1ull << lane) - 1ull));
    if ((uint64_t)base + __popcll(em) > a.row_cap) {
        if (lane == 0) atomicExch(&a.scalars[SC_OVERFLOW], 1ull);
        return;
    }
    if (!emit) return;
    uint32_t* out = a.rows[ct];
    const uint64_t cap = a.row_cap;
    out[0 * cap + row] = dp;
    out[1 * cap + row] = acc.nc;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        out[(2 + s) * cap + row] = acc.cc[s];
        out[(10 + s) * cap + row] = acc.bc[s];
        out[(18 + s) * cap + row] = acc.bq[s];
        out[(26 + s) * cap + row] = acc.bcf[s];
        out[(34 + s) * cap + row] = acc.bc[s] - acc.bcf[s];
    }
}

// ------------------------------------------------------------------------------------------------
// Wave kernel: each wavefront pulls units with <= CAPW entries from a queue.
constexpr int WAVES_PER_BLOCK = 4;
struct WaveLds {
    uint32_t rkey[CAPW], rev[CAPW], rmeta[CAPW];
    uint32_t tkey[HW], tcnt[HW];
    uint32_t pk[8 * 64];
    uint16_t order[CAPW];
};

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void k_pileup_wave(CountArgs a) {
    __shared__ WaveLds lds_all[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    WaveLds& L = lds_all[threadIdx.x >> 6];
    for (int i = lane; i < 8 * 64; i += 64) L.pk[i] = 0;
    const uint32_t n_ne = *a.n_ne;
    while (true) {
        uint32_t w = 0;
        if (lane == 0) w = (uint32_t)atomicAdd(&a.scalars[SC_QHEAD], 1ull);
        w = rl(w, 0);
        if (w >= n_ne) break;
        uint32_t u = a.ne_units[w];
        int n = (int)a.unit_cnt[u];
        if (n > CAPW) continue;                      // deep kernel's job
        uint32_t base = a.unit_off[u];
        int ct, tid; int32_t tstart;
        unit_geometry(a, u, ct, tid, tstart);
        lds_fence();
        for (int i = lane; i < n; i += 64) {
            Rec r = make_rec(a, a.entries[base + i], tstart);
            L.rkey[i] = r.key; L.rev[i] = r.ev_lo; L.rmeta[i] = r.meta;
        }
        lds_fence();
        group_by_cb<false, HW>(n, L.rkey, L.tkey, L.tcnt, L.order, lane, 64, nullptr);
        Acc acc; acc.init();
        walk(a, acc, L.order, L.rkey, L.rev, L.rmeta, 0, n, L.pk, lane);
        acc.finish(L.pk, lane);
        emit_unit(a, acc, w, ct, tid, tstart, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Deep kernel: one workgroup per unit with > CAPW entries.  Barcodes are split into passes of
// <= CAPB entries by a coarse histogram; each pass is staged, grouped and walked by 4 waves on
// run-aligned slices.  Counters are additive over disjoint barcode sets.
struct DeepLds {
    uint32_t rkey[CAPB], rev[CAPB], rmeta[CAPB];
    uint32_t tkey[HB], tcnt[HB];
    uint32_t hist[NBUCKET];
    uint32_t pk[DEEP_THREADS / 64][8 * 64];
    uint32_t acc[NCTR][64];
    uint32_t ormask[64];
    uint32_t wave_tot[DEEP_THREADS / 64];
    uint32_t pass_lo[NBUCKET + 1];     // pass p covers buckets [pass_lo[p], pass_lo[p+1])
    uint32_t n_pass, scount;
    uint16_t order[CAPB];
};

__global__ __launch_bounds__(DEEP_THREADS) void k_pileup_deep(CountArgs a) {
    __shared__ DeepLds L;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int NW = DEEP_THREADS / 64;
    const uint32_t n_deep = (uint32_t)a.scalars[SC_NDEEP];
    int shift = 0;
    while (((uint32_t)(a.n_cb - 1) >> shift) >= (uint32_t)NBUCKET) ++shift;

    for (uint32_t d = blockIdx.x; d < n_deep; d += gridDim.x) {
        uint32_t w = a.deep_list[d];
        uint32_t u = a.ne_units[w];
        const int n = (int)a.unit_cnt[u];
        const uint32_t base = a.unit_off[u];
        int ct, tid; int32_t tstart;
        unit_geometry(a, u, ct, tid, tstart);

        __syncthreads();
        for (int i = t; i < NBUCKET; i += DEEP_THREADS) L.hist[i] = 0;
        for (int i = t; i < 8 * 64; i += DEEP_THREADS) { for (int q = 0; q < NW; ++q) L.pk[q][i] = 0; }
        for (int i = t; i < NCTR * 64; i += DEEP_THREADS) (&L.acc[0][0])[i] = 0;
        __syncthreads();
        for (int i = t; i < n; i += DEEP_THREADS) atomicAdd(&L.hist[(a.entries[base + i].x & CB_MASK) >> shift], 1u);
        __syncthreads();
        if (t == 0) {
            // greedy grouping of consecutive buckets into passes of <= CAPB entries; a bucket that
            // alone exceeds CAPB becomes its own pass (handled per barcode value, stream mode).
            uint32_t np = 0, cur = 0; bool open = false;
            for (uint32_t b = 0; b < (uint32_t)NBUCKET; ++b) {
                uint32_t h = L.hist[b];
                if (h == 0) continue;
                if (!open || cur + h > (uint32_t)CAPB || h > (uint32_t)CAPB) { L.pass_lo[np++] = b; cur = 0; open = true; }
                cur += h;
                if (h > (uint32_t)CAPB) open = false;       // next non-empty bucket starts a new pass
            }
            L.pass_lo[np] = NBUCKET;
            L.n_pass = np;
        }
        __syncthreads();
        const uint32_t n_pass = L.n_pass;
        Acc acc; acc.init();

        for (uint32_t p = 0; p < n_pass; ++p) {
            const uint32_t b_lo = L.pass_lo[p];
            // the pass ends before the next pass's first bucket; buckets between are empty
            const uint32_t b_hi = L.pass_lo[p + 1];
            const bool overflow = L.hist[b_lo] > (uint32_t)CAPB;
            if (!overflow) {
                __syncthreads();
                if (t == 0) L.scount = 0;
                __syncthreads();
                for (int i = t; i < n; i += DEEP_THREADS) {
                    uint2 e = a.entries[base + i];
                    uint32_t b = (e.x & CB_MASK) >> shift;
                    if (b >= b_lo && b < b_hi) {
                        uint32_t slot = atomicAdd(&L.scount, 1u);
                        Rec r = make_rec(a, e, tstart);
                        L.rkey[slot] = r.key; L.rev[slot] = r.ev_lo; L.rmeta[slot] = r.meta;
                    }
                }
                __syncthreads();
                const int ns = (int)L.scount;
                group_by_cb<true, HB>(ns, L.rkey, L.tkey, L.tcnt, L.order, t, DEEP_THREADS, L.wave_tot);
                // run-aligned slice of this wave
                int j0 = (int)((int64_t)ns * wv / NW), j1 = (int)((int64_t)ns * (wv + 1) / NW);
                while (j0 > 0 && j0 < ns && (L.rkey[L.order[j0]] & CB_MASK) == (L.rkey[L.order[j0 - 1]] & CB_MASK)) ++j0;
                while (j1 > 0 && j1 < ns && (L.rkey[L.order[j1]] & CB_MASK) == (L.rkey[L.order[j1 - 1]] & CB_MASK)) ++j1;
                if (j0 > j1) j0 = j1;
                acc.close_run(); acc.prev_cb = KEY_INVALID;
                walk(a, acc, L.order, L.rkey, L.rev, L.rmeta, j0, j1, L.pk[wv], lane);
                acc.close_run(); acc.prev_cb = KEY_INVALID;
            } else {
                // stream mode: one barcode value at a time, no staging; the barcode's symbol masks of
                // the 4 waves are OR-ed before counting it as one cell.
                const uint32_t c_lo = b_lo << shift, c_hi = (b_lo + 1) << shift;
                for (uint32_t c = c_lo; c < c_hi && c < (uint32_t)a.n_cb; ++c) {
                    __syncthreads();
                    if (t < 64) L.ormask[t] = 0;
                    __syncthreads();
                    acc.close_run(); acc.prev_cb = KEY_INVALID;
                    uint32_t any = 0;
                    const int per = (n + NW - 1) / NW;
                    const int i0 = wv * per, i1 = (i0 + per < n) ? i0 + per : n;
                    for (int ib = i0; ib < i1; ib += 64) {
                        uint2 e = make_uint2(KEY_INVALID, 0);
                        if (ib + lane < i1) e = a.entries[base + ib + lane];
                        bool match = e.x != KEY_INVALID && (e.x & CB_MASK) == c;
                        Rec r{0, 0, 0};
                        if (match) r = make_rec(a, e, tstart);
                        unsigned long long mm = __ballot(match);
                        while (mm) {
                            int l = __ffsll((long long)mm) - 1; mm &= mm - 1;
                            uint32_t ks = rl(r.key, l), es = rl(r.ev_lo, l), ms = rl(r.meta, l);
                            uint32_t rel = (uint32_t)lane - ((ms >> 8) & 63u);
                            bool inr = rel <= ((ms >> 16) & 63u);
                            uint64_t addr = (((uint64_t)(ms & 0xffu)) << 32 | es) + rel;
                            uint32_t evv = inr ? (uint32_t)a.events[addr] : 0xffffu;
                            acc.add(evv, inr, ((ks >> 24) & 1u) ^ 1u, a.min_bq, L.pk[wv], lane);
                            any = 1;
                        }
                    }
                    (void)any;
                    if (acc.mask) atomicOr(&L.ormask[lane], acc.mask);
                    acc.mask = 0;
                    __syncthreads();
                    if (wv == 0) { acc.mask = L.ormask[lane]; acc.close_run(); }
                }
            }
        }
        acc.finish(L.pk[wv], lane);
        // reduce the 4 waves' accumulators
        __syncthreads();
        atomicAdd(&L.acc[0][lane], acc.nc);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            atomicAdd(&L.acc[1 + s][lane], acc.cc[s]);
            atomicAdd(&L.acc[9 + s][lane], acc.bc[s]);
            atomicAdd(&L.acc[17 + s][lane], acc.bq[s]);
            atomicAdd(&L.acc[25 + s][lane], acc.bcf[s]);
        }
        __syncthreads();
        if (wv == 0) {
            Acc tot; tot.init();
            tot.nc = L.acc[0][lane];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                tot.cc[s] = L.acc[1 + s][lane]; tot.bc[s] = L.acc[9 + s][lane];
                tot.bq[s] = L.acc[17 + s][lane]; tot.bcf[s] = L.acc[25 + s][lane];
            }
            emit_unit(a, tot, w, ct, tid, tstart, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct NonEmpty {
    const uint32_t* cnt;
    __host__ __device__ bool operator()(const uint32_t& i) const { return cnt[i] != 0; }
};

static int64_t host_scalar(lsg_ctx* c, int idx) {
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, c->d_scalars.as<unsigned long long>() + idx, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return (int64_t)v;
}

static void fill_args(lsg_ctx* c, const lsg_count_params* p, CountArgs& a) {
    a.n_reads = c->rd.n_reads; a.n_segs = c->rd.n_segs;
    a.read_tid = c->rd.read_tid; a.read_flag = c->rd.read_flag; a.read_mapq = c->rd.read_mapq; a.read_cb = c->rd.read_cb;
    a.seg_read = c->rd.seg_read; a.seg_start = c->rd.seg_start; a.seg_len = c->rd.seg_len; a.seg_ev_off = c->rd.seg_ev_off;
    a.events = c->rd.events;
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.ref_ptr = c->d_ref_ptrs.as<const uint8_t*>(); a.celltype_of = c->d_celltype_of.as<uint8_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.n_ct = c->n_ct;
    a.n_units = c->n_tiles * (uint32_t)c->n_ct;
    a.min_bq = p->min_bq; a.min_mq = p->min_mq; a.min_dp = p->min_dp; a.min_cc = p->min_cc;
    a.ignore_orphans = p->ignore_orphans; a.flag_exclude = p->flag_exclude;
    a.read_key = c->d_read_key.as<uint32_t>(); a.unit_cnt = c->d_unit_cnt.as<uint32_t>();
    a.unit_off = c->d_unit_off.as<uint32_t>(); a.unit_cursor = c->d_unit_fill.as<uint32_t>();
    a.entries = c->d_entries.as<uint2>();
    a.ne_units = c->d_ne_units.as<uint32_t>();
    a.n_ne = reinterpret_cast<const uint32_t*>(c->d_scalars.as<unsigned long long>() + 2);
    a.ne_mask = c->d_ne_mask.as<uint64_t>(); a.ne_rowbase = c->d_ne_rowbase.as<uint32_t>();
    a.deep_list = c->d_deep_list.as<uint32_t>();
    a.scalars = c->d_scalars.as<unsigned long long>();
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) a.rows[i] = c->d_rows[i].as<uint32_t>();
    a.row_cap = c->row_cap;
}

int run_count(lsg_ctx* c, const lsg_count_params* p) {
    if (c->n_contigs <= 0) { set_error("lsg_pileup_count: no contigs set"); return -2; }
    if (c->n_ct <= 0) { set_error("lsg_pileup_count: no barcodes set"); return -2; }
    if (!c->rd.events && c->rd.n_events > 0) { set_error("lsg_pileup_count: no reads loaded"); return -2; }
    for (int t = 0; t < c->n_contigs; ++t)
        if (!c->ref_ptr[t]) { set_error("lsg_pileup_count: reference of contig %d not loaded", t); return -2; }
    hipStream_t st = c->stream;
    const uint32_t n_units = c->n_tiles * (uint32_t)c->n_ct;
    const int64_t R = c->rd.n_reads, S = c->rd.n_segs;

    if (c->d_read_key.reserve((size_t)(R + 1) * 4)) return -1;
    if (c->d_unit_cnt.reserve(((size_t)n_units + 2) * 4)) return -1;
    if (c->d_unit_off.reserve(((size_t)n_units + 2) * 4)) return -1;
    if (c->d_unit_fill.reserve(((size_t)n_units + 2) * 4)) return -1;
    if (c->d_entries.reserve((size_t)(c->entries_upper + 1) * 8)) return -1;
    if (c->d_scalars.reserve(SC_COUNT * 8)) return -1;

    LSG_HIP(hipEventRecord(c->ev[0], st));
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, st));
    LSG_HIP(hipMemsetAsync(c->d_unit_cnt.p, 0, ((size_t)n_units + 1) * 4, st));

    CountArgs a{};
    fill_args(c, p, a);
    if (R > 0) hipLaunchKernelGGL(k_read_key, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, a);
    if (S > 0) hipLaunchKernelGGL(k_bin_segments<false>, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, a);

    // exclusive scan of unit counts (n_units + 1 elements so that off[n_units] = total entries)
    size_t tmp1 = 0, tmp2 = 0;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp1, a.unit_cnt, a.unit_off, (int)(n_units + 1), st));
    hipcub::CountingInputIterator<uint32_t> cnt_it(0);
    NonEmpty pred{a.unit_cnt};
    // the non-empty list can hold at most min(n_units, entries_upper) units
    uint64_t ne_cap = n_units < c->entries_upper ? n_units : c->entries_upper;
    if (c->d_ne_units.reserve((size_t)(ne_cap + 1) * 4)) return -1;
    if (c->d_ne_mask.reserve((size_t)(ne_cap + 1) * 8)) return -1;
    if (c->d_ne_rowbase.reserve((size_t)(ne_cap + 1) * 4)) return -1;
    if (c->d_deep_list.reserve((size_t)(c->entries_upper / (CAPW + 1) + 2) * 4)) return -1;
    fill_args(c, p, a);
    uint32_t* d_nsel = reinterpret_cast<uint32_t*>(c->d_scalars.as<unsigned long long>() + 2);
    LSG_HIP(hipcub::DeviceSelect::If(nullptr, tmp2, cnt_it, a.ne_units, d_nsel, (int)n_units, pred, st));
    if (c->d_cub_tmp.reserve((tmp1 > tmp2 ? tmp1 : tmp2) + 16)) return -1;
    size_t tmp = c->d_cub_tmp.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tmp, a.unit_cnt, a.unit_off, (int)(n_units + 1), st));
    tmp = c->d_cub_tmp.cap;
    LSG_HIP(hipcub::DeviceSelect::If(c->d_cub_tmp.p, tmp, cnt_it, a.ne_units, d_nsel, (int)n_units, pred, st));
    LSG_HIP(hipMemcpyAsync(c->d_unit_fill.p, c->d_unit_off.p, ((size_t)n_units + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (S > 0) hipLaunchKernelGGL(k_bin_segments<true>, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, a);

    // sizing read-back: number of non-empty units (grid of k_deep_list, row capacity)
    int64_t n_ne = host_scalar(c, 2) & 0xffffffffll;
    if (n_ne < 0) { set_error("lsg_pileup_count: scalar read-back failed"); return -1; }
    c->n_ne = (uint32_t)n_ne;
    uint64_t want_rows = (uint64_t)n_ne * TILE_W;
    if (p->min_dp > 0) {
        uint64_t by_depth = (uint64_t)c->rd.n_events / (uint64_t)p->min_dp + 64;
        if (by_depth < want_rows) want_rows = by_depth;
    }
    if (want_rows < 64) want_rows = 64;
    if (want_rows > c->row_cap) {
        for (int i = 0; i < c->n_ct; ++i)
            if (c->d_rows[i].reserve((size_t)want_rows * LSG_ROW_WORDS * 4)) return -1;
        c->row_cap = want_rows;
    }
    fill_args(c, p, a);
    LSG_HIP(hipEventRecord(c->ev[1], st));

    if (n_ne > 0) {
        hipLaunchKernelGGL(k_deep_list, dim3((unsigned)((n_ne + 255) / 256)), dim3(256), 0, st, a);
        hipDeviceProp_t prop;
        LSG_HIP(hipGetDeviceProperties(&prop, c->device));
        int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // deep units first (long-running), then the wave queue fills the machine around them
        hipLaunchKernelGGL(k_pileup_deep, dim3((unsigned)(cus * 2)), dim3(DEEP_THREADS), 0, st, a);
        uint64_t wave_blocks = ((uint64_t)n_ne + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        uint64_t max_blocks = (uint64_t)cus * 8;
        hipLaunchKernelGGL(k_pileup_wave, dim3((unsigned)(wave_blocks < max_blocks ? wave_blocks : max_blocks)),
                           dim3(WAVES_PER_BLOCK * 64), 0, st, a);
    }
    LSG_HIP(hipEventRecord(c->ev[2], st));
    LSG_HIP(hipGetLastError());

    unsigned long long sc[SC_COUNT];
    LSG_HIP(hipMemcpyAsync(sc, c->d_scalars.p, sizeof(sc), hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    if (sc[SC_OVERFLOW]) { set_error("lsg_pileup_count: row buffer overflow (cap %llu)", (unsigned long long)c->row_cap); return -3; }
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) c->n_rows[i] = (int64_t)sc[SC_ROWS + i];
    c->n_columns = (int64_t)sc[SC_COLS];
    c->n_deep = (uint32_t)sc[SC_NDEEP];
    c->stats.n_reads_admitted = (int64_t)sc[SC_READS];
    c->stats.n_segs_admitted = (int64_t)sc[SC_SEGS];
    c->stats.n_events_admitted = (int64_t)sc[SC_EVENTS];
    c->stats.n_units = n_ne;
    c->stats.n_deep_units = c->n_deep;
    uint32_t total_entries = 0;
    LSG_HIP(hipMemcpy(&total_entries, c->d_unit_off.as<uint32_t>() + n_units, 4, hipMemcpyDeviceToHost));
    c->stats.n_entries = total_entries;
    float ms = 0;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->stats.ms_bin = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->stats.ms_pileup = ms;
    LSG_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[2])); c->stats.ms_total = ms;
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
    uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= *a.n_ne) return;
    uint32_t u = a.ne_units[w];
    int uct, tid; int32_t tstart;
    unit_geometry(a, u, uct, tid, tstart);
    if (uct != ct) return;
    uint64_t em = a.ne_mask[w];
    if (!((em >> lane) & 1ull)) return;
    uint64_t src = (uint64_t)a.ne_rowbase[w] + __popcll(em & ((1ull << lane) - 1ull));
    uint64_t dst = (uint64_t)rowoff[w] + __popcll(em & ((1ull << lane) - 1ull));
    int64_t pos = (int64_t)tstart + lane;
    keys[dst] = ((int64_t)tid << 32) | pos;
    refs[dst] = a.ref_ptr[tid][pos];
    for (int k = 0; k < LSG_ROW_WORDS; ++k) counts[dst * LSG_ROW_WORDS + k] = a.rows[ct][(uint64_t)k * a.row_cap + src];
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
    size_t tmp = 0;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, cnt, off, (int)(n_ne + 1), st));
    if (c->d_cub_tmp.reserve(tmp + 16)) return -1;
    tmp = c->d_cub_tmp.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tmp, cnt, off, (int)(n_ne + 1), st));
    DevBuf dk, dr, dc;
    if (dk.reserve((size_t)n * 8) || dr.reserve((size_t)n) || dc.reserve((size_t)n * LSG_ROW_WORDS * 4)) { dk.release(); dr.release(); dc.release(); return -1; }
    uint64_t threads = (uint64_t)n_ne * 64;
    hipLaunchKernelGGL(k_export_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, ct, off,
                       dk.as<int64_t>(), dr.as<uint8_t>(), dc.as<uint32_t>());
    int rc = 0;
    if (hipMemcpyAsync(keys, dk.p, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(ref, dr.p, (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(counts, dc.p, (size_t)n * LSG_ROW_WORDS * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("lsg_fetch_counts: copy failed: %s", hipGetErrorString(hipGetLastError()));
        rc = -1;
    }
    dk.release(); dr.release(); dc.release();
    return rc;
}

// Upper bound of tile entries (sum over segments of tiles overlapped), computed once at load time.
__global__ void k_entries_upper(const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, unsigned long long* out) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (s < n_segs) {
        int32_t st = seg_start[s], ln = seg_len[s];
        if (st >= 0 && ln > 0) v = (unsigned long long)(((uint32_t)(st + ln - 1) >> 6) - ((uint32_t)st >> 6) + 1);
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

int compute_entries_upper(lsg_ctx* c) {
    if (c->d_scalars.reserve(SC_COUNT * 8)) return -1;
    LSG_HIP(hipMemsetAsync(c->d_scalars.p, 0, SC_COUNT * 8, c->stream));
    int64_t S = c->rd.n_segs;
    if (S > 0)
        hipLaunchKernelGGL(k_entries_upper, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream, c->rd.seg_start,
                           c->rd.seg_len, S, c->d_scalars.as<unsigned long long>());
    int64_t v = host_scalar(c, 0);
    if (v < 0) { set_error("lsg_load_reads: entries bound read-back failed"); return -1; }
    if ((uint64_t)v >= 0xFFFFFFF0ull) { set_error("lsg_load_reads: %lld tile entries exceed the 32-bit entry index; load the reads in windows", (long long)v); return -2; }
    c->entries_upper = (uint64_t)v;
    return 0;
}

} // namespace lsg
