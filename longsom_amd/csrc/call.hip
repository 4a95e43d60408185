// Outer join of the per-cell-type count rows + step-1 beta-binomial call, and position-set probes.
//
// Replaces
//   merge_cell_types_files     workflow/scripts/SNVCalling/MergeBaseCellCounts.py:116-204
//   variant_calling_step1      workflow/scripts/SNVCalling/BaseCellCalling.step1.py:19-476
//   build_dict / membership    workflow/scripts/SNVCalling/BaseCellCalling.step2.py:142-158,197-221
//
// One lane per merged site: the units (tile, cell type) of one tile are adjacent in the non-empty
// unit list, their 64-bit emit masks OR-ed give the tile's sites, and a site's row of cell type c is
// rowbase_c + popcount(mask_c below the lane).  The beta-binomial upper tail P(X >= k) that scipy's
// generic rv_discrete.sf evaluates as 1 - sum_{m<k} exp(logpmf(m)) (step1.py:196,201,329-330) is
// summed here in fp64 from the shorter side with the pmf ratio recurrence, re-anchored with lgamma
// every 1024 terms; p-values are stored as Python round(p, 4) * 1e4 (exact half-even on the binary
// value).  Arithmetic type f64; the text the host prints is identical whenever |p - tie| > ~1e-13.
//
// Device storage is compact (48 B per site + 56 B per (candidate site, cell type)); lsg_fetch_calls /
// lsg_export_calls expand it into the C-ABI's lsg_call records.
#include "lsg_ctx.h"
#include "call_rec.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

namespace lsg {

// counter words of the call stage; the ones the kernels allocate from while they run sit on cache lines of their own (atomics on one
// line serialise at ~90 per microsecond whichever of its words they name)
enum { CT_CAND = 0, CT_NCAND = 1, CT_HEADS = 4, CT_PASS = 6, CT_LIGHT = 16, CT_HEAVY = 32, CT_QTAIL = 48, CT_DEFER = 64, CT_WORDS = 80 };
struct CallArgs {
    const uint32_t* ne_units; const uint64_t* ne_mask; const uint32_t* ne_rowbase; const int2* ne_geom;
    uint32_t n_ne; int32_t n_ct;
    const uint32_t* rows[LSG_MAX_CELLTYPES]; uint64_t row_cap;
    const uint8_t* const* ref_ptr; const int64_t* contig_len;
    lsg_call_params p;
    double lgc0[2], lgcn[2];          // lgamma(a+b) - lgamma(b), lgamma(a+b) - lgamma(a) for (a1,b1), (a2,b2)
    uint32_t* pass_list;              // site indices of the PASS candidates (k_call_finish appends, counters[CT_PASS] counts; PASS_CAP slots)
    const int16_t* tail_table;        // [2][TAIL_ENTRIES] rounded tails of every (k <= n <= TAIL_NT), see k_tail_table
    uint32_t* site_cnt; uint32_t* site_off;
    SiteRec* sites; CandCt* cands; uint64_t cand_cap;
    struct TailTask* light; struct TailTask* heavy; uint64_t task_cap;
    unsigned long long* counters;     // CT_*: candidate blocks allocated, candidate sites (exact), light / heavy tail task slots, heads, ...
    uint32_t arena_waves;             // waves of k_call_gather (each owns chunk number `wave` of every arena list)
    const uint32_t* heads;            // units that are the first of a tile with at least one site
    uint64_t* head_ref;               // the head's contig's reference bases (address), prefetched beside its record
    uint32_t* head_recs;              // 16 words per head: contig length, first site index, tile start, tid, masks[4] (lo, hi), row bases[4]
    uint32_t* defer_list;             // sites with a tail still to be computed (k_call_gather appends, counters[CT_DEFER] counts): k_call_finish's work
    uint64_t defer_cap;
};

// log of the beta-binomial pmf at m (scipy betabinom._logpmf written with lgamma)
__device__ __noinline__ double bb_logpmf(double m, double n, double a, double b) {
    return lgamma(n + 1.0) - lgamma(m + 1.0) - lgamma(n - m + 1.0) + lgamma(m + a) + lgamma(n - m + b) - lgamma(n + a + b) +
           lgamma(a + b) - lgamma(a) - lgamma(b);
}

// P(X >= k) for X ~ BetaBinomial(n, a, b), integer k.  pm0 = pmf(0), shared by the alts of one cell type.
__device__ __noinline__ double bb_upper_tail(uint32_t k, uint32_t n, double a, double b, double pm0, double lgcn) {
    if (k == 0) return 1.0;
    if (k > n) return 0.0;                         // 1 - sum of the whole pmf; canonical 0.0
    const double dn = (double)n;
    if ((uint64_t)k <= (uint64_t)n - k + 1) {      // lower side is shorter: 1 - sum_{m<k} pmf(m)
        double sum = pm0, pm = pm0;
        for (uint32_t m = 1; m < k; ++m) {
            if ((m & 1023u) == 0) pm = exp(bb_logpmf((double)m, dn, a, b));
            else { const double mm = (double)(m - 1); pm *= (dn - mm) * (mm + a) / ((mm + 1.0) * (dn - mm - 1.0 + b)); }
            sum += pm;
        }
        return 1.0 - sum;
    }
    // upper side: sum_{m=k}^{n} pmf(m), descending from pmf(n) = G(n+a) G(a+b) / (G(n+a+b) G(a))
    double pm = exp(lgamma(dn + a) - lgamma(dn + a + b) + lgcn);
    double sum = pm;
    uint32_t cnt = 1;
    for (uint32_t m = n; m > k;) {
        --m;
        if ((cnt & 1023u) == 0) pm = exp(bb_logpmf((double)m, dn, a, b));
        else { const double mm = (double)m; pm *= (mm + 1.0) * (dn - mm - 1.0 + b) / ((dn - mm) * (mm + a)); }
        sum += pm;
        ++cnt;
    }
    return sum;
}
__device__ __noinline__ double bb_pm0(uint32_t n, double a, double b, double lgc0) {
    const double dn = (double)n;                   // pmf(0) = G(n+b) G(a+b) / (G(n+a+b) G(b))
    return exp(lgamma(dn + b) - lgamma(dn + a + b) + lgc0);
}

// Python round(x, 4) * 10^4 as an integer: half-even on the exact binary value of x.
__device__ __forceinline__ int32_t round4(double x) {
    if (!(x > 0.0)) return 0;                      // negative fp noise and -0.0 print as 0.0 (canonical)
    const double hi = x * 1e4;
    const double lo = fma(x, 1e4, -hi);
    double k = rint(hi);                           // half-even
    const double d = (hi - k) + lo;                // exact distance to k unless |lo| is absorbed (then irrelevant)
    if (d > 0.5) k += 1.0;
    else if (d < -0.5) k -= 1.0;
    else if (d == 0.5 && (hi - k) != 0.5) { if (fmod(k, 2.0) != 0.0) k += 1.0; }
    else if (d == -0.5 && (hi - k) != -0.5) { if (fmod(k, 2.0) != 0.0) k -= 1.0; }
    return (int32_t)k;
}

__device__ __forceinline__ int sym_of_ref(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'T' ? 2 : b == 'G' ? 3 : -1; }
__device__ __forceinline__ uint8_t base_of_sym(int s) { return s == 0 ? 'A' : s == 1 ? 'C' : s == 2 ? 'T' : 'G'; }

__global__ void k_site_count(CallArgs a) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > a.n_ne) return;
    uint32_t v = 0;
    if (w < a.n_ne) {
        const uint32_t tile = a.ne_units[w] / (uint32_t)a.n_ct;
        const bool head = w == 0 || a.ne_units[w - 1] / (uint32_t)a.n_ct != tile;
        if (head) {
            uint64_t m = 0;
            for (uint32_t q = w; q < a.n_ne && a.ne_units[q] / (uint32_t)a.n_ct == tile; ++q) m |= a.ne_mask[q];
            v = (uint32_t)__popcll(m);
        }
    }
    a.site_cnt[w] = v;
}

// ---- step 1 in three kernels ---------------------------------------------------------------------
//  k_call_gather  one lane per merged site: joins the cell types' rows, finds the alt candidates, the
//                 Rest_* sums and the homopolymer flags, writes the records WITHOUT p-values and emits
//                 one task (k, n, parameter set, destination) per beta-binomial tail;
//  k_call_tails   one thread per light task (<= 64 terms), one wavefront per heavy task (lanes sum
//                 strided chunks, each re-anchored by lgamma);
//  k_call_finish  the filter chains, which depend on the rounded p-values.
struct TailTask { uint32_t k, n; uint64_t dst; };     // dst = address of the int16 result | parameter set in bit 0

// Most tail requests have a small n (depth or cell count of one cell type at one site) and the same (k, n) pairs recur
// millions of times: all pairs k <= n <= TAIL_NT are evaluated once per parameter set (k_tail_table, with exactly the code a
// task of that pair would run, so the rounded values are the same bits) and looked up by k_call_gather instead of becoming tasks.
constexpr uint32_t PASS_CAP = 4096;      // PASS candidates are a few hundred per sample: list + in-block sort; more fall back to the scan
constexpr uint32_t TAIL_NT = 2047;
constexpr uint32_t TAIL_ENTRIES = (TAIL_NT + 1) * (TAIL_NT + 2) / 2;
__host__ __device__ __forceinline__ uint32_t tail_index(uint32_t k, uint32_t n) { return n * (n + 1) / 2 + k; }
__device__ __forceinline__ bool tail_in_table(uint32_t k, uint32_t n) { return n <= TAIL_NT && k <= n; }

__device__ __forceinline__ uint32_t tail_work(uint32_t k, uint32_t n) {
    if (k == 0 || k > n) return 0;
    const uint32_t up = n - k + 1;
    return k < up ? k : up;
}

// One 64-byte record per tile head: everything k_call_gather needs to know about the tile, so that a wave fetches it
// with ONE scalar load, a tile ahead of its use (the chain heads -> units -> masks -> row bases was 4 dependent round trips).
__global__ void k_head_recs(CallArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)a.counters[CT_HEADS]) return;
    const uint32_t w = a.heads[i];
    const uint32_t tile = a.ne_units[w] / (uint32_t)a.n_ct;
    uint64_t mask[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};
    uint32_t rbase[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};
    for (uint32_t q = w; q < a.n_ne && q < w + (uint32_t)a.n_ct && a.ne_units[q] / (uint32_t)a.n_ct == tile; ++q) {
        const int ct = (int)(a.ne_units[q] % (uint32_t)a.n_ct);
        mask[ct] = a.ne_mask[q]; rbase[ct] = a.ne_rowbase[q];
    }
    const int2 geom = a.ne_geom[w];
    a.head_ref[i] = (uint64_t)(uintptr_t)a.ref_ptr[geom.y & 0xffffff];
    uint32_t* r = a.head_recs + (uint64_t)i * 16;
    r[0] = (uint32_t)a.contig_len[geom.y & 0xffffff]; r[1] = a.site_off[w]; r[2] = (uint32_t)geom.x; r[3] = (uint32_t)geom.y;
    for (int ct = 0; ct < LSG_MAX_CELLTYPES; ++ct) { r[4 + 2 * ct] = (uint32_t)mask[ct]; r[5 + 2 * ct] = (uint32_t)(mask[ct] >> 32); r[12 + ct] = rbase[ct]; }
}

// Persistent waves: every wave walks tile heads (heads[]) with a stride and allocates candidate blocks and tail tasks
// from wave-private arenas refilled in chunks (a single counter word takes only ~90 atomics/us; per-site or even
// per-workgroup atomics would cap the kernel).  Task arenas are split across chunks so that only the LAST chunk of a
// wave has unused slots; those are written as null tasks (dst = 0) which the tail kernels skip.
constexpr int GATHER_WAVES = 4;
constexpr uint32_t CAND_CHUNK = 256, TASK_CHUNK = 64, HEAVY_CHUNK = 16;   // heavy tasks are rare and a null heavy slot costs a wave a memory round trip
// (127 registers per lane at two cell types: 4 waves per SIMD.  Capped at 96 / 80 registers the spills cost more than the fifth and sixth wave hide: 2.2 -> 2.5 / 3.1 ms)
template <int NCT>
__global__ __launch_bounds__(GATHER_WAVES * 64) __attribute__((amdgpu_waves_per_eu(NCT <= 2 ? 4 : 3))) void k_call_gather(CallArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));   // (uniform, and known to be: the head records below are scalar loads)
    const uint32_t n_waves = (uint32_t)(((uint64_t)gridDim.x * blockDim.x) >> 6);
    const uint32_t n_heads = (uint32_t)a.counters[CT_HEADS];
    // arenas: every wave starts with chunk number `wave` of each list (no storm of same-line atomics at launch); the
    // counters count the chunks taken AFTER those, so list positions and lengths are offset by n_waves chunks
    const uint32_t c_base = n_waves * CAND_CHUNK, l_base = n_waves * TASK_CHUNK, h_base = n_waves * HEAVY_CHUNK;
    uint32_t c_next = wave * CAND_CHUNK, c_end = c_next + CAND_CHUNK;                      // candidate blocks
    uint32_t l_next = wave * TASK_CHUNK, l_end = l_next + TASK_CHUNK, h_next = wave * HEAVY_CHUNK, h_end = h_next + HEAVY_CHUNK;  // light / heavy task slots
    uint32_t n_cand_exact = 0;
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    const __attribute__((address_space(4))) u32x16* H = (const __attribute__((address_space(4))) u32x16*)(uintptr_t)a.head_recs;
    const __attribute__((address_space(4))) uint64_t* HR = (const __attribute__((address_space(4))) uint64_t*)(uintptr_t)a.head_ref;
    u32x16 nxt = {};
    uint64_t ref_nxt = 0;
    if (wave < n_heads) { nxt = H[wave]; ref_nxt = HR[wave]; }
    for (uint32_t hi_ = wave; hi_ < n_heads; hi_ += n_waves) {
    const u32x16 rec = nxt;
    const uint8_t* ref = reinterpret_cast<const uint8_t*>((uintptr_t)ref_nxt);
    if (hi_ + n_waves < n_heads) { nxt = H[hi_ + n_waves]; ref_nxt = HR[hi_ + n_waves]; }      // the next tile's record travels while this tile is worked on
    uint64_t mask[NCT];
    uint32_t rbase[NCT];
    uint64_t any = 0;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { mask[ct] = (uint64_t)rec[4 + 2 * ct] | ((uint64_t)rec[5 + 2 * ct] << 32); rbase[ct] = rec[12 + ct]; any |= mask[ct]; }
    const bool site = (any >> lane) & 1ull;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t idx = (uint64_t)rec[1] + __popcll(any & below);
    const int2 geom = make_int2((int)rec[2], (int)rec[3]);
    const int tid = geom.y & 0xffffff;
    const int64_t pos = (int64_t)geom.x + lane;
    // Every load of the tile is issued here, before anything is waited for (they used to sit in branches and loops that each
    // ended in a wait: reference base, one cell type's rows, the other's, then the ten context bytes one round trip at a time).
    // The reference: lane l reads the base of its own position, lanes 0..9 also the five bases either side of the tile; the
    // homopolymer context of every lane (step1.py:95-107) is cut from those 74 bases below without another load.
    const int64_t clen = (int64_t)rec[0];
    const int64_t gx = (int64_t)geom.x;
    const int64_t xpos = lane < 5 ? gx - 5 + lane : gx + 59 + lane;          // lanes 5..9: gx + 64 .. gx + 68
    const bool r0_ok = pos < clen, r1_ok = lane < 10 && xpos >= 0 && xpos < clen;
    const uint8_t r0_raw = ref[r0_ok ? pos : 0], r1_raw = ref[r1_ok ? xpos : 0];
    // rows: planes 0..15 = DP, NC, CC[0..7], BC[0..5] in four 16-byte loads (quad k at 256 words from quad k - 1); a narrow row
    // (small unit) holds the same planes as 16-bit values, 8 bytes a quad at 128 words: read with the same 16-byte loads (the
    // upper half is not used).  A lane without a row reads row `lane` of block 0, so that no load sits in a branch.
    struct __attribute__((packed, aligned(4))) Quad { uint32_t x, y, z, w; };
    Quad q[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const bool has_row = site && ((mask[ct] >> lane) & 1ull);
        const uint64_t row = has_row ? (uint64_t)(rbase[ct] & ~ROW_NARROW) + __popcll(mask[ct] & below) : (uint64_t)lane;
        const bool narrow = has_row && (rbase[ct] & ROW_NARROW);
        const char* R = reinterpret_cast<const char*>(a.rows[ct] + (row >> 6) * ROW_BLOCK_WORDS) + (narrow ? (row & 63) * 8 : (row & 63) * 16);
        const uint32_t stride = narrow ? 512u : 1024u;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[ct][k] = *reinterpret_cast<const Quad*>(R + k * stride);
    }
    const uint8_t r0 = r0_ok ? r0_raw : (uint8_t)0, r1 = r1_ok ? r1_raw : (uint8_t)0;
    const uint8_t refb = site ? r0 : (uint8_t)'N';
    const int rsym = sym_of_ref(refb);
    const lsg_call_params& P = a.p;
    const int order[4] = {0, 1, 3, 2};              // letter order A < C < G < T over classes (A,C,T,G) = (0,1,2,3)
    uint32_t v_dp[NCT], v_nc[NCT], v_cc[NCT][6], v_bc[NCT][6];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        uint4 q0 = make_uint4(q[ct][0].x, q[ct][0].y, q[ct][0].z, q[ct][0].w), q1 = make_uint4(q[ct][1].x, q[ct][1].y, q[ct][1].z, q[ct][1].w);
        uint4 q2 = make_uint4(q[ct][2].x, q[ct][2].y, q[ct][2].z, q[ct][2].w), q3 = make_uint4(q[ct][3].x, q[ct][3].y, q[ct][3].z, q[ct][3].w);
        if (rbase[ct] & ROW_NARROW) {
            q0 = make_uint4(q0.x & 0xffffu, q0.x >> 16, q0.y & 0xffffu, q0.y >> 16); q1 = make_uint4(q1.x & 0xffffu, q1.x >> 16, q1.y & 0xffffu, q1.y >> 16);
            q2 = make_uint4(q2.x & 0xffffu, q2.x >> 16, q2.y & 0xffffu, q2.y >> 16); q3 = make_uint4(q3.x & 0xffffu, q3.x >> 16, q3.y & 0xffffu, q3.y >> 16);
        }
        v_dp[ct] = q0.x; v_nc[ct] = q0.y;
        v_cc[ct][0] = q0.z; v_cc[ct][1] = q0.w; v_cc[ct][2] = q1.x; v_cc[ct][3] = q1.y; v_cc[ct][4] = q1.z; v_cc[ct][5] = q1.w;
        v_bc[ct][0] = q2.z; v_bc[ct][1] = q2.w; v_bc[ct][2] = q3.x; v_bc[ct][3] = q3.y; v_bc[ct][4] = q3.z; v_bc[ct][5] = q3.w;
    }
    // the context: E bit i says "the base at gx - 5 + i equals the one before it" (wave-uniform, 74 bits); a lane's upstream
    // five are positions pos-5..pos-1 (comparisons E[lane+1..lane+4]), its downstream five pos+1..pos+5 (E[lane+7..lane+10])
    const uint32_t prev0 = (uint32_t)__shfl_up((int)r0, 1), prev1 = (uint32_t)__shfl_up((int)r1, 1), next0 = (uint32_t)__shfl_down((int)r0, 1);
    const uint64_t eq0 = __ballot(lane > 0 && r0 == (uint8_t)prev0), eq1 = __ballot(lane > 0 && lane != 5 && lane < 10 && r1 == (uint8_t)prev1);
    const uint32_t r1_4 = (uint32_t)__builtin_amdgcn_readlane((int)r1, 4), r1_5 = (uint32_t)__builtin_amdgcn_readlane((int)r1, 5);
    const uint32_t r0_0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, 0), r0_63 = (uint32_t)__builtin_amdgcn_readlane((int)r0, 63);
    // X = E >> 1 as (xlo, xhi): bit j of X is E[j + 1]
    const uint64_t e_lo = (eq1 & 0x1eull) | ((uint64_t)(r0_0 == r1_4) << 5) | (eq0 << 5);                       // E[0..63]: eq0 bit l -> i = l + 5
    const uint64_t e_hi = (eq0 >> 59) | ((uint64_t)(r1_5 == r0_63) << 5) | (((eq1 >> 6) & 0xfull) << 6);        // E[64..73]: i - 64
    const uint64_t xlo = (e_lo >> 1) | (e_hi << 63), xhi = e_hi >> 1;
    const uint32_t xw = (uint32_t)((xlo >> lane) | (lane ? xhi << (64 - lane) : 0ull));                         // bit j = E[lane + 1 + j]
    const bool have_ctx = pos >= 5;
    int down_len = (int)((clen - (pos + 1)) < 5 ? (clen - (pos + 1)) : 5);
    if (down_len < 0) down_len = 0;
    const uint32_t ue = xw & 15u;                                             // up[1]==up[0], .., up[4]==up[3]
    const uint32_t de = down_len > 0 ? (xw >> 6) & ((1u << (down_len - 1)) - 1u) : 0u;   // down[1]==down[0], .. (only inside the contig)
    auto longest = [](uint32_t m) { const uint32_t m2 = m & (m >> 1), m3 = m2 & (m >> 2), m4 = m3 & (m >> 3); return (int)((m != 0) + (m2 != 0) + (m3 != 0) + (m4 != 0)); };
    const int up_best = 1 + longest(ue), up_run = 1 + (int)__builtin_clz((((~ue) & 15u) << 28) | (1u << 27));    // run that ends in up[4]
    const int dn_best = 1 + longest(de), dn_run = 1 + (int)__builtin_ctz((~de) | 16u);                         // run that starts at down[0]
    const uint8_t up4 = lane ? (uint8_t)prev0 : (uint8_t)r1_4, dn0 = lane < 63 ? (uint8_t)next0 : (uint8_t)r1_5;
    // pass 1: candidate?  the Rest_* sums; does any tail of the site lie outside the table?  (selects, no branches: the lanes of a
    // tile disagree at every one of these tests)
    bool has_any = false, off_table = false;
    uint32_t n_light = 0, n_heavy = 0;
    int32_t s_alts_bc = 0, s_alts_cc = 0, s_dp = 0, s_nc = 0;
    if (site) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const uint32_t dp = v_dp[ct], nc = v_nc[ct];
            const bool ok = ((mask[ct] >> lane) & 1ull) && (int)dp >= P.min_cov && (int)nc >= P.min_cells;
            s_dp += ok ? (int32_t)dp : 0; s_nc += ok ? (int32_t)nc : 0;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const bool nr = ok && s != rsym;
                const uint32_t b = nr ? v_bc[ct][s] : 0u, c = nr ? v_cc[ct][s] : 0u;
                const bool alt = s < 4 && b > 0;
                has_any |= alt;
                s_alts_bc += alt ? 0 : (int32_t)b; s_alts_cc += alt ? 0 : (int32_t)c;
                s_dp -= alt ? (int32_t)b : 0; s_nc -= alt ? (int32_t)c : 0;
                off_table |= alt && !(tail_in_table(b, dp) && tail_in_table(c, nc));
            }
        }
        if (s_alts_bc > 0) {
            off_table |= s_dp >= 0 && !tail_in_table((uint32_t)s_alts_bc, (uint32_t)s_dp);
            off_table |= s_nc >= 0 && s_alts_cc >= 0 && !tail_in_table((uint32_t)s_alts_cc, (uint32_t)s_nc);
        }
    }
    if (__ballot(off_table)) {                                  // (the deepest sites only) how many tasks, of which kind
        if (site) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (!((mask[ct] >> lane) & 1ull)) continue;
                const uint32_t dp = v_dp[ct], nc = v_nc[ct];
                if (!((int)dp >= P.min_cov && (int)nc >= P.min_cells)) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t b = v_bc[ct][s], c = v_cc[ct][s];
                    if (s == rsym || b == 0) continue;
                    if (!tail_in_table(b, dp)) { if (tail_work(b, dp) > 64) ++n_heavy; else ++n_light; }
                    if (!tail_in_table(c, nc)) { if (tail_work(c, nc) > 64) ++n_heavy; else ++n_light; }
                }
            }
            if (s_alts_bc > 0) {
                if (s_dp >= 0 && !tail_in_table((uint32_t)s_alts_bc, (uint32_t)s_dp)) { if (tail_work((uint32_t)s_alts_bc, (uint32_t)s_dp) > 64) ++n_heavy; else ++n_light; }
                if (s_nc >= 0 && s_alts_cc >= 0 && !tail_in_table((uint32_t)s_alts_cc, (uint32_t)s_nc)) { if (tail_work((uint32_t)s_alts_cc, (uint32_t)s_nc) > 64) ++n_heavy; else ++n_light; }
            }
        }
    }
    const unsigned long long cm = __ballot(has_any);
    // wave-inclusive prefix sums of the task counts
    uint32_t pl = n_light, ph = n_heavy, tot_l = 0, tot_h = 0;
    if (__ballot((n_light | n_heavy) != 0u)) {                 // (most tiles ask the table only)
        for (int o = 1; o < 64; o <<= 1) { const uint32_t vl = __shfl_up(pl, o), vh = __shfl_up(ph, o); if (lane >= o) { pl += vl; ph += vh; } }
        tot_l = __shfl(pl, 63); tot_h = __shfl(ph, 63);
    }
    // wave-private arenas
    const uint32_t n_c = (uint32_t)__popcll(cm);
    n_cand_exact += n_c;
    if (c_next + n_c > c_end) {                              // candidate blocks may leave holes: they are reached through sr.cand only
        uint32_t nb = 0;
        if (lane == 0) nb = (uint32_t)atomicAdd(&a.counters[CT_CAND], (unsigned long long)CAND_CHUNK) + c_base;
        c_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb); c_end = c_next + CAND_CHUNK;
    }
    const uint32_t cbase = c_next; c_next += n_c;
    // task slots: positions [pos, pos + tot) of the wave's stream; a request that does not fit continues in a new chunk
    uint32_t l_old = l_next, l_room = l_end - l_next, l_new = 0;
    if (tot_l > l_room) {
        uint32_t nb = 0;
        if (lane == 0) nb = (uint32_t)atomicAdd(&a.counters[CT_LIGHT], (unsigned long long)TASK_CHUNK * ((tot_l - l_room + TASK_CHUNK - 1) / TASK_CHUNK)) + l_base;
        l_new = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        l_next = l_new + (tot_l - l_room); l_end = l_new + TASK_CHUNK * ((tot_l - l_room + TASK_CHUNK - 1) / TASK_CHUNK);
    } else l_next += tot_l;
    uint32_t h_old = h_next, h_room = h_end - h_next, h_new = 0;
    if (tot_h > h_room) {
        uint32_t nb = 0;
        if (lane == 0) nb = (uint32_t)atomicAdd(&a.counters[CT_HEAVY], (unsigned long long)HEAVY_CHUNK * ((tot_h - h_room + HEAVY_CHUNK - 1) / HEAVY_CHUNK)) + h_base;
        h_new = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        h_next = h_new + (tot_h - h_room); h_end = h_new + HEAVY_CHUNK * ((tot_h - h_room + HEAVY_CHUNK - 1) / HEAVY_CHUNK);
    } else h_next += tot_h;
    if (!site) continue;
    const uint32_t cand = cbase + (uint32_t)__popcll(cm & below);
    uint32_t li = pl - n_light, hi = ph - n_heavy;           // this lane's first slot, as an offset into the wave's request
    // loc = the field of the record still being assembled in registers (stored whole afterwards), dst = the same field in memory
    // every tail the table can answer is read here, all of them at once (looked up where they are used, each was a round trip of
    // its own: up to 18 in a row); a pair outside the table reads entry 0 and becomes a task below
    const int16_t* TT = a.tail_table;
    auto table_at = [&](uint32_t k, uint32_t n, int set) { return TT[tail_in_table(k, n) ? (uint32_t)set * TAIL_ENTRIES + tail_index(k, n) : 0u]; };
    int16_t t_bc[NCT][4], t_cc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { t_bc[ct][s] = 0; t_cc[ct][s] = 0; }
    }
    if (cm) {                                                  // (a tile without a candidate has only the noise tails to ask for)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { t_bc[ct][s] = table_at(v_bc[ct][s], v_dp[ct], 0); t_cc[ct][s] = table_at(v_cc[ct][s], v_nc[ct], 1); }
        }
    }
    const int16_t t_nb = table_at((uint32_t)s_alts_bc, (uint32_t)s_dp, 0), t_nc = table_at((uint32_t)s_alts_cc, (uint32_t)s_nc, 1);
    auto emit_task = [&](uint32_t k, uint32_t n, int set, int16_t* dst, int16_t* loc, int16_t from_table) {
        if (tail_in_table(k, n)) { *loc = from_table; return; }
        TailTask t; t.k = k; t.n = n; t.dst = (uint64_t)dst | (uint64_t)set;
        if (tail_work(k, n) > 64) { const uint32_t p = hi < h_room ? h_old + hi : h_new + (hi - h_room); if (p < a.task_cap) a.heavy[p] = t; ++hi; }
        else { const uint32_t p = li < l_room ? l_old + li : l_new + (li - l_room); if (p < a.task_cap) a.light[p] = t; ++li; }
    };

    SiteRec sr;
    sr.key = ((int64_t)tid << 32) | pos; sr.ref = refb; sr.present = 0; sr.considered = 0; sr.has_cand = 0;
    sr.pad[0] = sr.pad[1] = sr.pad[2] = 0; sr.pad2 = 0; sr.cand = has_any ? cand : 0xFFFFFFFFu;
    sr.site_filter = 0;
    int n_considered = 0;
    bool alts_differ = false;
    uint32_t first_altset = 0; bool have_first = false;
    int lc_up = 0, lc_down = 0;
    SiteRec* srp = &a.sites[idx];
    // every tail of this site came out of the table (n <= TAIL_NT: all but the deepest sites): its filter chains are finished right here;
    // the others are listed for k_call_finish, which runs when the tail kernels have written their values
    const bool all_known = (n_light | n_heavy) == 0u;
    int f_pass = 0, f_nonsig = 0, f_with = 0; bool f_multi = false;

#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        CandCt cd;
        cd.n_alt = 0; cd.ct_filter = 0; cd.pad[0] = cd.pad[1] = 0;
#pragma unroll
        for (int q = 0; q < LSG_CALL_MAX_ALT; ++q) { cd.alt[q] = 0; cd.alt_bc[q] = 0; cd.alt_cc[q] = 0; cd.p_bc[q] = 0; cd.p_cc[q] = 0; }
        CandCt* cdp = has_any && cand < a.cand_cap ? &a.cands[(uint64_t)cand * NCT + ct] : nullptr;
        if ((mask[ct] >> lane) & 1ull) {
            sr.present |= (uint8_t)(1u << ct);
            const uint32_t dp = v_dp[ct], nc = v_nc[ct];
            if ((int)dp >= P.min_cov && (int)nc >= P.min_cells) {                 // step1.py:174
                sr.considered |= (uint8_t)(1u << ct);
                ++n_considered;
                // candidates: every observed A/C/T/G alt (:195-208), in letter order
                int na = 0; uint32_t altset = 0;
#pragma unroll
                for (int oi = 0; oi < 4; ++oi) {
                    const int s = order[oi];
                    const uint32_t b = v_bc[ct][s];
                    if (s == rsym || b == 0) continue;
                    const uint32_t c = v_cc[ct][s];
                    if (na < LSG_CALL_MAX_ALT) {
                        cd.alt[na] = (uint8_t)s; cd.alt_bc[na] = b; cd.alt_cc[na] = c;
                        if (cdp) { emit_task(b, dp, 0, &cdp->p_bc[na], &cd.p_bc[na], t_bc[ct][s]); emit_task(c, nc, 1, &cdp->p_cc[na], &cd.p_cc[na], t_cc[ct][s]); }
                    }
                    ++na; altset |= 1u << s;
                }
                cd.n_alt = (uint8_t)na;
                if (na > 0) {
                    sr.has_cand |= (uint8_t)(1u << ct);
                    if (!have_first) { first_altset = altset; have_first = true; } else if (altset != first_altset) alts_differ = true;
                    // homopolymer runs including the alt string "A" or "A|C|.." (:511-529): only the first /
                    // last letter of the string touches the context
                    if (have_ctx) {
                        int first_s = -1, last_s = -1;
                        for (int oi = 0; oi < 4; ++oi) { const int s = order[oi]; if ((altset >> s) & 1u) { if (first_s < 0) first_s = s; last_s = s; } }
                        const int u = base_of_sym(first_s) == up4 ? up_run + 1 : 1;             // upstream: longestRun(up + x)
                        const int ub = u > up_best ? u : up_best;
                        lc_up = ub > lc_up ? ub : lc_up;
                        int db = 1;                                                               // downstream: longestRun(x + down)
                        if (down_len > 0) { const int d = base_of_sym(last_s) == dn0 ? dn_run + 1 : 1; db = d > dn_best ? d : dn_best; }
                        lc_down = db > lc_down ? db : lc_down;
                    }
                }
            }
        }
        if (all_known && cd.n_alt > 0) {                                               // the per-cell-type chain (:263-277; k_call_finish's, same order)
            int32_t min_pbc = 100000, min_pcc = 100000;
#pragma unroll
            for (int q = 0; q < LSG_CALL_MAX_ALT; ++q) if (q < (int)cd.n_alt) { min_pbc = cd.p_bc[q] < min_pbc ? cd.p_bc[q] : min_pbc; min_pcc = cd.p_cc[q] < min_pcc ? cd.p_cc[q] : min_pcc; }
            uint8_t f;
            if (min_pbc >= 500 || min_pcc >= 500) f = LSG_CF_NONSIG;
            else if ((min_pbc > 10 && min_pbc < 500) || (min_pcc > 10 && min_pcc < 500)) f = LSG_CF_LOWSIG;
            else if (cd.n_alt > 1) f = LSG_CF_MULTI;
            else if ((int)cd.alt_cc[0] < P.min_ac_cells) f = LSG_CF_LOW_CELLS;
            else if ((int)cd.alt_bc[0] < P.min_ac_reads) f = LSG_CF_LOW_READS;
            else f = LSG_CF_PASS;
            cd.ct_filter = f;
            ++f_with; f_pass += f == LSG_CF_PASS; f_nonsig += f == LSG_CF_NONSIG; f_multi |= f == LSG_CF_MULTI;
        }
        if (cdp) *cdp = cd;
    }
    sr.cell_types_min = (uint8_t)n_considered;
    sr.sum_alts_bc = s_alts_bc; sr.sum_dp = s_dp; sr.sum_alts_cc = s_alts_cc; sr.sum_nc = s_nc;
    // noise tails (:328-337 / :426-435); a negative n is outside scipy's support -> nan
    sr.noise_p_bc = -1; sr.noise_p_cc = -1;
    if (s_alts_bc > 0) {
        sr.noise_p_bc = -2; sr.noise_p_cc = -2;
        if (s_dp >= 0) emit_task((uint32_t)s_alts_bc, (uint32_t)s_dp, 0, &srp->noise_p_bc, &sr.noise_p_bc, t_nb);
        if (s_nc >= 0 && s_alts_cc >= 0) emit_task((uint32_t)s_alts_cc, (uint32_t)s_nc, 1, &srp->noise_p_cc, &sr.noise_p_cc, t_nc);
    }
    // flags that do not depend on p-values; k_call_finish adds the rest
    uint32_t sf = 0;
    if (sr.has_cand) {
        sf |= LSG_SF_CANDIDATE;
        if (alts_differ) sf |= LSG_SF_MULTI_ALLELIC;                           // :313
        if (n_considered < P.min_cell_types) sf |= LSG_SF_MIN_CELL_TYPES;      // :318
        if (have_ctx && lc_up >= 4) sf |= LSG_SF_LC_UP;                        // :347-354
        if (have_ctx && lc_down >= 4) sf |= LSG_SF_LC_DOWN;
    }
    bool pass_site = false;
    if (all_known) {                                                           // the site's chain (k_call_finish's)
        const int32_t npb = sr.noise_p_bc, npc = sr.noise_p_cc;
        if (sr.has_cand) {
            if (f_pass > P.max_cell_types) sf |= LSG_SF_MULTIPLE_CELL_TYPES;       // :309
            if (f_multi) sf |= LSG_SF_MULTI_ALLELIC;                               // :314
            if (f_with - f_pass - f_nonsig > 0) sf |= LSG_SF_CELL_TYPE_NOISE;      // :322
            if (s_alts_bc > 0 && ((npb >= 0 && npb < 500) || (npc >= 0 && npc < 500))) sf |= LSG_SF_NOISY_SITE;   // :342
            pass_site = sf == (uint32_t)LSG_SF_CANDIDATE && f_pass > 0;
        } else if (s_alts_bc > 0 && ((npb >= 0 && npb < 10) || (npc >= 0 && npc < 10))) sf |= LSG_SF_NOISY_SITE;   // :440-442
    }
    sr.site_filter = sf;
    *srp = sr;
    if (pass_site) {                                                           // (a few hundred per sample)
        const unsigned long long q = atomicAdd(&a.counters[CT_PASS], 1ull);
        if (q < PASS_CAP) a.pass_list[q] = (uint32_t)idx;
    }
    const unsigned long long dm = __ballot(!all_known);
    if (dm) {
        unsigned long long q0 = 0;
        if (lane == __ffsll((long long)dm) - 1) q0 = atomicAdd(&a.counters[CT_DEFER], (unsigned long long)__popcll(dm));
        q0 = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(q0 >> 32), __ffsll((long long)dm) - 1) << 32) | (uint32_t)__shfl((int)(uint32_t)q0, __ffsll((long long)dm) - 1);
        const unsigned long long q = q0 + __popcll(dm & below);
        if (!all_known && q < a.defer_cap) a.defer_list[q] = (uint32_t)idx;
    }
    }
    // unused tail of the wave's last task chunks: null tasks
    TailTask nul; nul.k = 0; nul.n = 0; nul.dst = 0;
    for (uint32_t p = l_next + (uint32_t)lane; p < l_end; p += 64) if (p < a.task_cap) a.light[p] = nul;
    for (uint32_t p = h_next + (uint32_t)lane; p < h_end; p += 64) if (p < a.task_cap) a.heavy[p] = nul;
    if (lane == 0 && n_cand_exact) atomicAdd(&a.counters[CT_NCAND], (unsigned long long)n_cand_exact);
}

__device__ __forceinline__ double tail_of_task(const TailTask& t, const CallArgs& a) {
    const int set = (int)(t.dst & 1ull);
    const double al = set ? a.p.alpha2 : a.p.alpha1, be = set ? a.p.beta2 : a.p.beta1;
    return bb_upper_tail(t.k, t.n, al, be, bb_pm0(t.n, al, be, a.lgc0[set]), a.lgcn[set]);
}

__global__ __launch_bounds__(256) void k_call_tails(CallArgs a) {
    const uint64_t n_all = a.counters[CT_LIGHT] + (uint64_t)a.arena_waves * TASK_CHUNK;      // the waves' first chunks + the chunks taken later
    const uint64_t n = n_all < a.task_cap ? n_all : a.task_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const TailTask t = a.light[i];
        if (!(t.dst & ~1ull)) continue;                        // unused arena slot
        *reinterpret_cast<int16_t*>(t.dst & ~1ull) = (int16_t)round4(tail_of_task(t, a));
    }
}

// G lanes, one tail (G = 64: the whole wavefront; G = 8: eight tails side by side, each lane group with its own k, n, set): lane g
// of a group sums the terms [g*chunk, (g+1)*chunk) of the shorter side; the value is valid on the group's first lane.  Every lane
// pays one log-pmf (nine lgamma) to anchor its chunk, so a tail of a few hundred terms is cheaper on 8 lanes than on 64.
template <int G>
__device__ __forceinline__ double group_tail(uint32_t k, uint32_t n, int set, const CallArgs& a, int lane, bool live = true) {
    const double al = set ? a.p.alpha2 : a.p.alpha1, be = set ? a.p.beta2 : a.p.beta1;
    const double dn = (double)n;
    const bool lower = (uint64_t)k <= (uint64_t)n - k + 1;
    const uint32_t m_lo = lower ? 0u : k, m_hi = live ? (lower ? k : n + 1) : m_lo;        // terms m in [m_lo, m_hi)
    const uint32_t cnt = m_hi - m_lo, chunk = (cnt + G - 1) / G;
    const uint32_t b = m_lo + (uint32_t)(lane & (G - 1)) * chunk;
    const uint32_t e = b + chunk < m_hi ? b + chunk : m_hi;
    double sum = 0.0;
    if (b < e) {
        double pm = exp(bb_logpmf((double)b, dn, al, be));
        sum = pm;
        for (uint32_t m = b + 1; m < e; ++m) {
            if (((m - b) & 1023u) == 0) pm = exp(bb_logpmf((double)m, dn, al, be));
            else { const double mm = (double)(m - 1); pm *= (dn - mm) * (mm + al) / ((mm + 1.0) * (dn - mm - 1.0 + be)); }
            sum += pm;
        }
    }
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, G);
    return lower ? 1.0 - sum : sum;
}
__device__ __forceinline__ uint32_t tail_terms(uint32_t k, uint32_t n) { return (uint64_t)k <= (uint64_t)n - k + 1 ? k : n + 1 - k; }
constexpr uint32_t GROUP8_MAX_TERMS = 1024;       // above this a tail gets the whole wavefront

// every (k, n) with k <= n <= TAIL_NT for both parameter sets: pairs a light task would take by one thread each, pairs a heavy
// task would take by one wavefront each (the same functions the task kernels call)
__global__ __launch_bounds__(256) void k_tail_table(CallArgs a, int16_t* table) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_thr = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t i = tid; i < 2ull * TAIL_ENTRIES; i += n_thr) {
        const int set = i >= TAIL_ENTRIES;
        const uint32_t idx = (uint32_t)(i - (set ? TAIL_ENTRIES : 0));
        uint32_t n = (uint32_t)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
        while (tail_index(0, n + 1) <= idx) ++n;
        while (tail_index(0, n) > idx) --n;
        const uint32_t k = idx - tail_index(0, n);
        if (tail_work(k, n) > 64) continue;
        TailTask t; t.k = k; t.n = n; t.dst = (uint64_t)set;
        table[i] = (int16_t)round4(tail_of_task(t, a));
    }
    for (uint64_t i = tid >> 3; i < 2ull * TAIL_ENTRIES; i += n_thr >> 3) {          // n <= TAIL_NT: at most 256 terms, 8 lanes each
        const int set = i >= TAIL_ENTRIES;
        const uint32_t idx = (uint32_t)(i - (set ? TAIL_ENTRIES : 0));
        uint32_t n = (uint32_t)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
        while (tail_index(0, n + 1) <= idx) ++n;
        while (tail_index(0, n) > idx) --n;
        const uint32_t k = idx - tail_index(0, n);
        const bool heavy = tail_work(k, n) > 64;
        const double tail = group_tail<8>(k, n, set, a, lane, heavy);
        if (heavy && (lane & 7) == 0) table[i] = (int16_t)round4(tail);
    }
}

// heavy tasks: eight lanes or one wavefront each
__global__ __launch_bounds__(256) void k_call_tails_heavy(CallArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_all = a.counters[CT_HEAVY] + (uint64_t)a.arena_waves * HEAVY_CHUNK;
    const uint64_t n_tasks = n_all < a.task_cap ? n_all : a.task_cap;
    // tasks differ by three orders of magnitude in their number of terms and most arena slots are empty: a wave takes 16
    // slots at a time off a queue (its first batch is its own index: a small job never touches the queue word), one slot per
    // lane, and works through the live ones
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t batch = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_batches = (n_tasks + 15) / 16;
    while (batch < n_batches) {
        // the 16 slots of a batch lie n_batches apart: live slots come in dense runs (one gather wave's chunk, one deep region)
        // and would otherwise all land on the same few waves
        const uint64_t i = (uint64_t)lane * n_batches + batch;
        TailTask t; t.k = 0; t.n = 0; t.dst = 0;
        if (lane < 16 && i < n_tasks) t = a.heavy[i];
        const bool is_live = (t.dst & ~1ull) != 0;
        const bool is_big = is_live && tail_terms(t.k, t.n) > GROUP8_MAX_TERMS;
        // tails of up to GROUP8_MAX_TERMS terms: eight at a time, lane group g takes slot 8 * round + g of the batch
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const int l = 8 * round + (lane >> 3);
            const uint32_t k = (uint32_t)__shfl((int)t.k, l), n = (uint32_t)__shfl((int)t.n, l);
            const uint64_t dst = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(t.dst >> 32), l) << 32) | (uint32_t)__shfl((int)(uint32_t)t.dst, l);
            const bool mine = (dst & ~1ull) != 0 && tail_terms(k, n) <= GROUP8_MAX_TERMS;
            if (__ballot(mine) == 0ull) continue;
            const double tail = group_tail<8>(k, n, (int)(dst & 1ull), a, lane, mine);
            if (mine && (lane & 7) == 0) *reinterpret_cast<int16_t*>(dst & ~1ull) = (int16_t)round4(tail);
        }
        unsigned long long big = __ballot(is_big);
        while (big) {                                                  // the long ones: the whole wavefront each
            const int l = __ffsll((long long)big) - 1; big &= big - 1;
            const uint32_t k = (uint32_t)__shfl((int)t.k, l), n = (uint32_t)__shfl((int)t.n, l);
            const uint64_t dst = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(t.dst >> 32), l) << 32) | (uint32_t)__shfl((int)(uint32_t)t.dst, l);
            const double tail = group_tail<64>(k, n, (int)(dst & 1ull), a, lane);
            if (lane == 0) *reinterpret_cast<int16_t*>(dst & ~1ull) = (int16_t)round4(tail);
        }
        unsigned long long nb = 0;
        if (lane == 0) nb = atomicAdd(&a.counters[CT_QTAIL], 1ull) + n_waves;
        batch = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
    }
}

__global__ void k_call_finish(CallArgs a, uint32_t n_sites) {
    const unsigned long long n_def = a.counters[CT_DEFER] < a.defer_cap ? a.counters[CT_DEFER] : a.defer_cap;
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_def; j += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t i = a.defer_list[j];
    if (i >= n_sites) continue;
    SiteRec s = a.sites[i];
    const lsg_call_params& P = a.p;
    const int32_t npb = s.noise_p_bc, npc = s.noise_p_cc;
    const bool bc_lt05 = npb >= 0 && npb < 500, cc_lt05 = npc >= 0 && npc < 500;
    const bool bc_lt001 = npb >= 0 && npb < 10, cc_lt001 = npc >= 0 && npc < 10;
    uint32_t sf = s.site_filter;
    if (s.has_cand) {
        int n_pass = 0, n_nonsig = 0, n_with = 0; bool any_multi = false;
        for (int ct = 0; ct < a.n_ct; ++ct) {
            if (!((s.has_cand >> ct) & 1u)) continue;
            CandCt* d = &a.cands[(uint64_t)s.cand * a.n_ct + ct];
            const int na = d->n_alt;
            int32_t min_pbc = 100000, min_pcc = 100000;
            for (int q = 0; q < na && q < LSG_CALL_MAX_ALT; ++q) { min_pbc = d->p_bc[q] < min_pbc ? d->p_bc[q] : min_pbc; min_pcc = d->p_cc[q] < min_pcc ? d->p_cc[q] : min_pcc; }
            uint8_t f;                                                         // per-cell-type chain (:263-277), rounded values
            if (min_pbc >= 500 || min_pcc >= 500) f = LSG_CF_NONSIG;
            else if ((min_pbc > 10 && min_pbc < 500) || (min_pcc > 10 && min_pcc < 500)) f = LSG_CF_LOWSIG;
            else if (na > 1) f = LSG_CF_MULTI;
            else if ((int)d->alt_cc[0] < P.min_ac_cells) f = LSG_CF_LOW_CELLS;
            else if ((int)d->alt_bc[0] < P.min_ac_reads) f = LSG_CF_LOW_READS;
            else f = LSG_CF_PASS;
            d->ct_filter = f;
            ++n_with; n_pass += f == LSG_CF_PASS; n_nonsig += f == LSG_CF_NONSIG; any_multi |= f == LSG_CF_MULTI;
        }
        if (n_pass > P.max_cell_types) sf |= LSG_SF_MULTIPLE_CELL_TYPES;       // :309
        if (any_multi) sf |= LSG_SF_MULTI_ALLELIC;                             // :314
        if (n_with - n_pass - n_nonsig > 0) sf |= LSG_SF_CELL_TYPE_NOISE;      // :322
        if (s.sum_alts_bc > 0 && (bc_lt05 || cc_lt05)) sf |= LSG_SF_NOISY_SITE;   // :342
        if (sf == (uint32_t)LSG_SF_CANDIDATE && n_pass > 0) {                   // = keep_site(kind 2): what step 3 can keep
            const unsigned long long idx = atomicAdd(&a.counters[CT_PASS], 1ull);
            if (idx < PASS_CAP) a.pass_list[idx] = i;
        }
    } else if (s.sum_alts_bc > 0 && (bc_lt001 || cc_lt001)) {
        sf |= LSG_SF_NOISY_SITE;                                               // :440-442
    }
    if (sf != s.site_filter) a.sites[i].site_filter = sf;
    }
}

// Expansion of compact records into the C-ABI's lsg_call; kind selects rows (see lsg_export_calls).
__device__ __forceinline__ bool keep_site(const SiteRec& s, const CandCt* cands, int n_ct, int kind) {
    if (kind == 0) return true;
    if (kind == 1) return s.site_filter != 0;
    if (s.site_filter != (uint32_t)LSG_SF_CANDIDATE) return false;
    bool pass = false;
    for (int ct = 0; ct < n_ct; ++ct) pass |= cands[(uint64_t)s.cand * n_ct + ct].ct_filter == LSG_CF_PASS;
    return pass;
}
__global__ void k_flag_keep(const SiteRec* sites, const CandCt* cands, int n_ct, int64_t n, int kind, uint32_t* keep) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) keep[i] = (i < n && keep_site(sites[i], cands, n_ct, kind)) ? 1u : 0u;
}
__device__ __forceinline__ void expand_site(const SiteRec& s, const CandCt* cands, int n_ct, const uint8_t* const* ref_ptr, const int64_t* contig_len,
                                            lsg_call* dst) {
    lsg_call c;
    memset(&c, 0, sizeof(c));
    c.key = s.key; c.ref = s.ref; c.present = s.present; c.considered = s.considered; c.has_cand = s.has_cand;
    c.site_filter = s.site_filter; c.cell_types_min = s.cell_types_min;
    c.sum_alts_bc = s.sum_alts_bc; c.sum_dp = s.sum_dp; c.sum_alts_cc = s.sum_alts_cc; c.sum_nc = s.sum_nc;
    c.noise_p_bc = s.noise_p_bc; c.noise_p_cc = s.noise_p_cc;
    if (s.cand != 0xFFFFFFFFu) {
        for (int ct = 0; ct < n_ct; ++ct) {
            const CandCt& d = cands[(uint64_t)s.cand * n_ct + ct];
            c.n_alt[ct] = d.n_alt; c.ct_filter[ct] = d.ct_filter;
            for (int q = 0; q < LSG_CALL_MAX_ALT; ++q) {
                c.alt[ct][q] = d.alt[q]; c.alt_bc[ct][q] = d.alt_bc[q]; c.alt_cc[ct][q] = d.alt_cc[q];
                c.p_bc[ct][q] = d.p_bc[q]; c.p_cc[ct][q] = d.p_cc[q];
            }
        }
    }
    const int tid = (int)(s.key >> 32);
    const int64_t pos = s.key & 0xffffffffll;
    if (pos >= 5) {
        const uint8_t* ref = ref_ptr[tid];
        for (int q = 0; q < 5; ++q) c.up_ctx[q] = ref[pos - 5 + q];
        for (int q = 0; q < 5 && pos + 1 + q < contig_len[tid]; ++q) c.down_ctx[q] = ref[pos + 1 + q];
    }
    *dst = c;
}
__global__ void k_expand(const SiteRec* sites, const CandCt* cands, int n_ct, int64_t n, const uint32_t* keep, const uint32_t* off,
                         const uint8_t* const* ref_ptr, const int64_t* contig_len, lsg_call* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    expand_site(sites[i], cands, n_ct, ref_ptr, contig_len, out + off[i]);
}
// PASS candidates from the list k_call_finish wrote (append order): ONE workgroup sorts the <= PASS_CAP site indices back into
// genomic order (bitonic, LDS) and expands them
__global__ __launch_bounds__(1024) void k_expand_list(const SiteRec* sites, const CandCt* cands, int n_ct, const uint32_t* list, uint32_t n,
                                                      const uint8_t* const* ref_ptr, const int64_t* contig_len, lsg_call* out) {
    __shared__ uint32_t v[PASS_CAP];
    const uint32_t t = threadIdx.x;
    uint32_t m = 1; while (m < n) m <<= 1;
    for (uint32_t i = t; i < m; i += 1024) v[i] = i < n ? list[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < m; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t x = v[i], y = v[l];
                    if (((i & k) == 0) == (x > y)) { v[i] = y; v[l] = x; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = t; i < n; i += 1024) expand_site(sites[v[i]], cands, n_ct, ref_ptr, contig_len, out + i);
}

__global__ void k_probe(const int64_t* set, int64_t n_set, const int64_t* keys, int64_t n, uint8_t* hits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t key = keys[i];
    int64_t lo = 0, hi = n_set;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (set[mid] < key) lo = mid + 1; else hi = mid; }
    hits[i] = (lo < n_set && set[lo] == key) ? 1 : 0;
}

// round(betabinom.sf(k - 0.001, n, a, b), 4) * 1e4 for a batch of (k, n) pairs (lsg_betabinom_sf4)
__global__ void k_sf4_batch(const uint32_t* k, const uint32_t* n, int64_t items, double al, double be, double lgc0, double lgcn, int32_t* out, double* raw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (int64_t)gridDim.x * blockDim.x) {
        const double p = bb_upper_tail(k[i], n[i], al, be, bb_pm0(n[i], al, be, lgc0), lgcn);
        out[i] = round4(p);
        if (raw) raw[i] = p;
    }
}

int run_sf4(lsg_ctx* c, int64_t items, const uint32_t* k, const uint32_t* n, double al, double be, int32_t* out, double* raw) {
    if (items < 0 || (items > 0 && (!k || !n || !out))) { set_error("lsg_betabinom_sf4: bad arguments"); return -2; }
    if (!(al > 0.0) || !(be > 0.0)) { set_error("lsg_betabinom_sf4: alpha and beta must be positive"); return -2; }
    if (items == 0) return 0;
    hipStream_t st = c->stream;
    DevBuf dk, dn, dout, draw;
    auto done = [&](int rc) { dk.release(); dn.release(); dout.release(); draw.release(); return rc; };
    if (dk.reserve((size_t)items * 4) || dn.reserve((size_t)items * 4) || dout.reserve((size_t)items * 4) || (raw && draw.reserve((size_t)items * 8))) return done(-1);
    if (hipMemcpyAsync(dk.p, k, (size_t)items * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dn.p, n, (size_t)items * 4, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("lsg_betabinom_sf4: upload failed"); return done(-1); }
    unsigned g = (unsigned)((items + 255) / 256); if (g > (unsigned)(c->n_cus * 16)) g = (unsigned)(c->n_cus * 16);
    hipLaunchKernelGGL(k_sf4_batch, dim3(g), dim3(256), 0, st, dk.as<uint32_t>(), dn.as<uint32_t>(), items, al, be,
                       lgamma(al + be) - lgamma(be), lgamma(al + be) - lgamma(al), dout.as<int32_t>(), raw ? draw.as<double>() : (double*)nullptr);
    if (raw && hipMemcpyAsync(raw, draw.p, (size_t)items * 8, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("lsg_betabinom_sf4: failed"); return done(-1); }
    if (hipMemcpyAsync(out, dout.p, (size_t)items * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("lsg_betabinom_sf4: failed"); return done(-1); }
    return done(0);
}

struct HasSites {
    const uint32_t* cnt;
    __host__ __device__ bool operator()(const uint32_t& w) const { return cnt[w] != 0; }
};

static int run_call_sized(lsg_ctx* c, const lsg_call_params* p, uint32_t tasks_per_site);
// The tail tasks' buffers are sized for ONE task per merged site (C2 makes 0.08: the table answers the rest; 8 per site, the old
// size, was 6 GB of a fresh handle's first call); a sample that needs more is called again with the bound no site can exceed.
int run_call(lsg_ctx* c, const lsg_call_params* p) {
    int rc = run_call_sized(c, p, c->call_tasks_per_site);
    if (rc == -3) {
        c->call_tasks_per_site = (uint32_t)(8 * c->n_ct + 2);
        rc = run_call_sized(c, p, c->call_tasks_per_site);
    }
    return rc;
}
static int run_call_sized(lsg_ctx* c, const lsg_call_params* p, uint32_t tasks_per_site) {
    if (!c->counted) { set_error("lsg_call_step1: call lsg_pileup_count first"); return -2; }
    for (int t = 0; t < c->n_contigs; ++t)          // (the kernels read the sites' reference bases and contexts: installed count rows come without a check of their own)
        if (!c->ref_ptr[t]) { set_error("lsg_call_step1: reference of contig %d not loaded", t); return -2; }
    hipStream_t st = c->stream;
    const uint32_t n_ne = c->n_ne;
    c->n_sites = 0; c->n_cand = 0; c->n_pass = -1;
    if (n_ne == 0) { c->called = true; return 0; }
    if (c->d_site_off.reserve((size_t)(n_ne + 2) * 12 + 128 + CT_WORDS * 8 + (size_t)n_ne * 72 + 64)) return -1;     // site_cnt, site_off, counters, heads, head records, their reference pointers
    CallArgs a{};
    a.ne_units = c->d_ne_units.as<uint32_t>(); a.ne_mask = c->d_ne_mask.as<uint64_t>(); a.ne_rowbase = c->d_ne_rowbase.as<uint32_t>();
    a.ne_geom = c->ws[WS_NE_GEOM].as<int2>();
    a.n_ne = n_ne; a.n_ct = c->n_ct;
    for (int i = 0; i < LSG_MAX_CELLTYPES; ++i) a.rows[i] = c->d_rows[i].as<uint32_t>();
    for (int i = 0; i < c->n_ct; ++i) if (!a.rows[i] || c->d_rows[i].cap < ROW_BLOCK_WORDS * 4) { set_error("lsg_call_step1: a cell type has no row block"); return -2; }   // (k_call_gather reads block 0 for lanes without a row)
    a.row_cap = c->row_cap;
    a.ref_ptr = c->d_ref_ptrs.as<const uint8_t*>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.p = *p;
    a.lgc0[0] = lgamma(p->alpha1 + p->beta1) - lgamma(p->beta1); a.lgcn[0] = lgamma(p->alpha1 + p->beta1) - lgamma(p->alpha1);
    a.lgc0[1] = lgamma(p->alpha2 + p->beta2) - lgamma(p->beta2); a.lgcn[1] = lgamma(p->alpha2 + p->beta2) - lgamma(p->alpha2);
    {   // small-n tails, once per parameter set
        const double key[4] = {p->alpha1, p->beta1, p->alpha2, p->beta2};
        if (!c->tail_table_valid || memcmp(key, c->tail_table_key, sizeof key) != 0) {
            if (c->d_tail_table.reserve((size_t)2 * TAIL_ENTRIES * 2)) return -1;
            hipLaunchKernelGGL(k_tail_table, dim3((unsigned)(c->n_cus * 8)), dim3(256), 0, st, a, c->d_tail_table.as<int16_t>());
            memcpy(c->tail_table_key, key, sizeof key); c->tail_table_valid = true;
        }
        a.tail_table = c->d_tail_table.as<int16_t>();
    }
    if (c->d_pass_list.reserve((size_t)PASS_CAP * 4)) return -1;
    a.pass_list = c->d_pass_list.as<uint32_t>();
    a.site_cnt = c->d_site_off.as<uint32_t>();
    a.site_off = a.site_cnt + (n_ne + 2);
    a.counters = reinterpret_cast<unsigned long long*>(a.site_off + (n_ne + 2));   // 2*(n_ne+2) words: 8-byte aligned
    uint32_t* heads = reinterpret_cast<uint32_t*>(a.counters + CT_WORDS);
    a.heads = heads;
    a.head_recs = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(heads + n_ne) + 63) & ~(uintptr_t)63);
    a.head_ref = reinterpret_cast<uint64_t*>(a.head_recs + (size_t)n_ne * 16);
    LSG_HIP(hipMemsetAsync(a.counters, 0, CT_WORDS * 8, st));
    hipLaunchKernelGGL(k_site_count, dim3((n_ne + 256) / 256), dim3(256), 0, st, a);
    size_t tb = 0;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, a.site_cnt, a.site_off, (int)(n_ne + 1), st));
    if (c->d_cub_tmp.reserve(tb + 256)) return -1;
    tb = c->d_cub_tmp.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb, a.site_cnt, a.site_off, (int)(n_ne + 1), st));
    {   // the tile heads with at least one site, in order
        HasSites pred{a.site_cnt};
        hipcub::CountingInputIterator<uint32_t> it(0);
        uint32_t* d_nh = reinterpret_cast<uint32_t*>(a.counters + CT_HEADS);
        size_t tb2 = 0;
        LSG_HIP(hipcub::DeviceSelect::If(nullptr, tb2, it, heads, d_nh, (int)n_ne, pred, st));
        if (c->d_cub_tmp.reserve(tb2 + 256)) return -1;
        tb2 = c->d_cub_tmp.cap;
        LSG_HIP(hipcub::DeviceSelect::If(c->d_cub_tmp.p, tb2, it, heads, d_nh, (int)n_ne, pred, st));
    }
    hipLaunchKernelGGL(k_head_recs, dim3((n_ne + 255) / 256), dim3(256), 0, st, a);
    LSG_HIP(hipMemcpyAsync(c->h_pin, a.site_off + n_ne, 4, hipMemcpyDeviceToHost, st));     // pinned landing zone
    LSG_HIP(hipStreamSynchronize(st));
    const uint32_t n_sites = *reinterpret_cast<const uint32_t*>(c->h_pin);
    if (n_sites > 0) {
        if (c->d_calls.reserve((size_t)n_sites * sizeof(SiteRec))) return -1;
        const unsigned gather_grid = (unsigned)(c->n_cus * 4);
        const uint64_t gather_waves = (uint64_t)gather_grid * GATHER_WAVES;
        if (c->ws[WS_CALL_CANDS].reserve(((size_t)n_sites + (size_t)2 * gather_waves * CAND_CHUNK) * sizeof(CandCt) * (size_t)c->n_ct)) return -1;   // every site could be a candidate + arena slack
        a.arena_waves = (uint32_t)gather_waves;
        a.sites = c->d_calls.as<SiteRec>(); a.cands = c->ws[WS_CALL_CANDS].as<CandCt>(); a.cand_cap = n_sites + 2 * gather_waves * CAND_CHUNK;
        // at most 2 tails per alt (<= 4 alts) per cell type + 2 noise tails per site
        a.task_cap = (uint64_t)n_sites * (uint64_t)(8 * c->n_ct + 2);
        if (a.task_cap > (uint64_t)n_sites * tasks_per_site + 1024) a.task_cap = (uint64_t)n_sites * tasks_per_site + 1024;      // (checked below: run_call sizes again when it was too small)
        if (a.task_cap > 0x7fffffffull) a.task_cap = 0x7fffffffull;
        a.task_cap += gather_waves * TASK_CHUNK * 3;                                  // arena slack (first chunks + last partial chunks)
        if (c->ws[WS_CALL_TASKS].reserve((size_t)a.task_cap * sizeof(TailTask) * 2)) return -1;
        a.light = c->ws[WS_CALL_TASKS].as<TailTask>(); a.heavy = a.light + a.task_cap;
        a.defer_cap = n_sites;                                                        // (a site is listed once)
        if (c->d_defer_list.reserve((size_t)a.defer_cap * 4 + 64)) return -1;
        a.defer_list = c->d_defer_list.as<uint32_t>();
        switch (c->n_ct) {                                                            // the cell-type loops are compile-time
            case 1: hipLaunchKernelGGL(k_call_gather<1>, dim3(gather_grid), dim3(GATHER_WAVES * 64), 0, st, a); break;
            case 2: hipLaunchKernelGGL(k_call_gather<2>, dim3(gather_grid), dim3(GATHER_WAVES * 64), 0, st, a); break;
            case 3: hipLaunchKernelGGL(k_call_gather<3>, dim3(gather_grid), dim3(GATHER_WAVES * 64), 0, st, a); break;
            default: hipLaunchKernelGGL(k_call_gather<4>, dim3(gather_grid), dim3(GATHER_WAVES * 64), 0, st, a); break;
        }
        hipLaunchKernelGGL(k_call_tails, dim3((unsigned)(c->n_cus * 16)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_call_tails_heavy, dim3((unsigned)(c->n_cus * 8)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_call_finish, dim3((unsigned)(c->n_cus * 2)), dim3(256), 0, st, a, n_sites);        // (the listed sites only: C2: 294 k of 23.9 M)
        LSG_HIP(hipGetLastError());
    }
    unsigned long long cnt4[CT_WORDS];
    LSG_HIP(hipMemcpyAsync(c->h_pin, a.counters, CT_WORDS * 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    memcpy(cnt4, c->h_pin, CT_WORDS * 8);
    if (getenv("LSG_TIMING")) fprintf(stderr, "[lsg] call: %u sites in %llu tiles, light task slots %llu (+%llu of the first chunks), heavy %llu (+%llu), sites that waited for a tail %llu, PASS %llu\n", n_sites, cnt4[CT_HEADS],
                                      cnt4[CT_LIGHT], (unsigned long long)a.arena_waves * TASK_CHUNK, cnt4[CT_HEAVY], (unsigned long long)a.arena_waves * HEAVY_CHUNK, cnt4[CT_DEFER], cnt4[CT_PASS]);
    c->n_pass = (int64_t)cnt4[CT_PASS];
    const unsigned long long cand = cnt4[CT_NCAND];
    if (n_sites > 0 && (cnt4[CT_LIGHT] + (uint64_t)a.arena_waves * TASK_CHUNK > a.task_cap || cnt4[CT_HEAVY] + (uint64_t)a.arena_waves * HEAVY_CHUNK > a.task_cap)) { set_error("lsg_call_step1: tail task buffer too small (%llu/%llu tasks)", cnt4[CT_LIGHT], cnt4[CT_HEAVY]); return -3; }
    c->n_sites = n_sites; c->n_cand = (int64_t)cand;
    c->called = true;
    return 0;
}

// Compacts + expands the selected call records (genomic order kept) into a DEVICE buffer of lsg_call.
int run_select_calls(lsg_ctx* c, int kind, lsg_call* dst_device, int64_t capacity, int64_t* n_out) {
    if (!c->called) { set_error("lsg_export_calls: call lsg_call_step1 first"); return -2; }
    hipStream_t st = c->stream;
    const int64_t n = c->n_sites;
    if (n_out) *n_out = 0;
    if (n == 0) return 0;
    if (kind == 2 && c->n_pass >= 0 && c->n_pass <= (int64_t)PASS_CAP && !getenv("LSG_NO_PASS_LIST")) {
        // the call stage left the PASS candidates' site indices and their number behind: no scan over the sites, no count read
        const int64_t k = c->n_pass;
        if (n_out) *n_out = k;
        if (!dst_device || k == 0) return 0;
        if (k > capacity) { set_error("lsg_export_calls: capacity %lld < %lld rows", (long long)capacity, (long long)k); return -2; }
        hipLaunchKernelGGL(k_expand_list, dim3(1), dim3(1024), 0, st, c->d_calls.as<SiteRec>(), c->ws[WS_CALL_CANDS].as<CandCt>(), c->n_ct,
                           c->d_pass_list.as<uint32_t>(), (uint32_t)k, c->d_ref_ptrs.as<const uint8_t*>(), c->d_contig_len.as<int64_t>(), dst_device);
        LSG_HIP(hipGetLastError());
        LSG_HIP(hipStreamSynchronize(st));
        return 0;
    }
    DevBuf& flags = c->ws[WS_CALL_FLAGS];
    if (flags.reserve((size_t)(n + 2) * 8)) return -1;
    uint32_t* keep = flags.as<uint32_t>();
    uint32_t* off = keep + (n + 2);
    const SiteRec* sites = c->d_calls.as<SiteRec>();
    const CandCt* cands = c->ws[WS_CALL_CANDS].as<CandCt>();
    hipLaunchKernelGGL(k_flag_keep, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, sites, cands, c->n_ct, n, kind, keep);
    size_t tb = 0;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, keep, off, (int)(n + 1), st));
    if (c->d_cub_tmp.reserve(tb + 256)) return -1;
    tb = c->d_cub_tmp.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(c->d_cub_tmp.p, tb, keep, off, (int)(n + 1), st));
    LSG_HIP(hipMemcpyAsync(c->h_pin, off + n, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    const uint32_t k = *reinterpret_cast<const uint32_t*>(c->h_pin);
    if (n_out) *n_out = k;
    if (!dst_device || k == 0) return 0;
    if ((int64_t)k > capacity) { set_error("lsg_export_calls: capacity %lld < %lld rows", (long long)capacity, (long long)k); return -2; }
    hipLaunchKernelGGL(k_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sites, cands, c->n_ct, n, keep, off,
                       c->d_ref_ptrs.as<const uint8_t*>(), c->d_contig_len.as<int64_t>(), dst_device);
    LSG_HIP(hipGetLastError());
    LSG_HIP(hipStreamSynchronize(st));
    return 0;
}

int run_fetch_calls(lsg_ctx* c, lsg_call* out, int64_t capacity, int candidates_only, int64_t* n_out) {
    if (!c->called) { set_error("lsg_fetch_calls: call lsg_call_step1 first"); return -2; }
    hipStream_t st = c->stream;
    if (n_out) *n_out = 0;
    if (c->n_sites == 0) return 0;
    int64_t k = 0;
    int rc = run_select_calls(c, candidates_only, nullptr, 0, &k);
    if (rc) return rc;
    if (k > capacity) { set_error("lsg_fetch_calls: capacity %lld < %lld rows", (long long)capacity, (long long)k); return -2; }
    if (k > 0) {
        DevBuf& sel = c->ws[WS_CALL_SEL];
        if (sel.reserve((size_t)k * sizeof(lsg_call))) return -1;
        rc = run_select_calls(c, candidates_only, sel.as<lsg_call>(), k, &k);
        if (rc) return rc;
        LSG_HIP(hipMemcpyAsync(out, sel.p, (size_t)k * sizeof(lsg_call), hipMemcpyDeviceToHost, st));
        LSG_HIP(hipStreamSynchronize(st));
    }
    if (n_out) *n_out = k;
    return 0;
}

int run_probe(lsg_ctx* c, int kind, const int64_t* keys, int64_t n, uint8_t* hits, int on_device) {
    PosSet& s = c->posset[kind];
    hipStream_t st = c->stream;
    if (n == 0) return 0;
    DevBuf dk, dh;
    const int64_t* k = keys; uint8_t* h = hits;
    int rc = 0;
    if (!on_device) {
        if (dk.reserve((size_t)n * 8) || dh.reserve((size_t)n)) { dk.release(); dh.release(); return -1; }
        if (hipMemcpyAsync(dk.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, st) != hipSuccess) rc = -1;
        k = dk.as<int64_t>(); h = dh.as<uint8_t>();
    }
    if (!rc) {
        hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.keys.as<int64_t>(), s.n, k, n, h);
        if (!on_device && hipMemcpyAsync(hits, dh.p, (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) rc = -1;
        if (hipStreamSynchronize(st) != hipSuccess) rc = -1;
    }
    if (rc) set_error("lsg_probe_posset: %s", hipGetErrorString(hipGetLastError()));
    dk.release(); dh.release();
    return rc;
}

} // namespace lsg
