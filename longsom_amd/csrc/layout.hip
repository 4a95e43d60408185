// The pileup's depth cap (bam.pileup(..., max_depth), BaseCellCounter.py:191): the bound that says whether it can fire at all, and
// htslib's rule itself when it can.
#include "lsg_ctx.h"
#include <algorithm>
#include <vector>
#include <hipcub/hipcub.hpp>

namespace lsg {

// +1 at the tile a read's first segment starts in, -1 past the tile its last segment ends in; reads of cell type ct
// (celltype_of == nullptr: every read with a known barcode).  The reads of a deep gene start and end in the same few tiles and a
// word takes ~90 atomics per microsecond, so a workgroup first merges its marks in an LDS hash and issues one global atomic per
// distinct tile.
constexpr int SPAN_THREADS = 256, SPAN_H = 1024;
__global__ __launch_bounds__(SPAN_THREADS) void k_span_marks(const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs,
                             const int32_t* read_tid, const int32_t* read_cb, const uint8_t* celltype_of, int32_t n_cb, int32_t ct,
                             const uint32_t* tile_base, int32_t n_contigs, int32_t* diff) {
    __shared__ uint32_t hkey[SPAN_H];
    __shared__ int32_t hval[SPAN_H];
    for (int i = threadIdx.x; i < SPAN_H; i += SPAN_THREADS) { hkey[i] = 0xFFFFFFFFu; hval[i] = 0; }
    __syncthreads();
    auto mark = [&](uint32_t t, int32_t v) {
        uint32_t h = (t * 2654435761u) >> 22;
        while (true) {
            const uint32_t prev = atomicCAS(&hkey[h], 0xFFFFFFFFu, t);
            if (prev == 0xFFFFFFFFu || prev == t) break;
            h = (h + 1) & (SPAN_H - 1);
        }
        atomicAdd(&hval[h], v);
    };
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_segs) {
        const uint32_t r = seg_read[s];
        const int32_t tid = read_tid[r];
        const int32_t cb = read_cb[r];
        bool ok = !(cb < 0 || tid < 0 || tid >= n_contigs);
        if (ok && celltype_of && (cb >= n_cb || celltype_of[cb] != ct)) ok = false;
        const bool first = s == 0 || seg_read[s - 1] != r, last = s + 1 == n_segs || seg_read[s + 1] != r;
        if (ok && (first || last)) {
            // the read is still buffered while the column AFTER its last one is entered (freed by that column's sweep): span end inclusive
            int64_t st = seg_start[s], en = st + (seg_len[s] > 0 ? seg_len[s] : 0);
            const uint32_t tb = tile_base[tid], te = tile_base[tid + 1];
            if (te > tb) {
                if (st < 0) st = 0;
                if (en < 0) en = 0;
                // both marks are clamped into the contig so that every +1 has its -1 (a bound must never under-count)
                if (first) { uint32_t t = tb + (uint32_t)(st >> 6); if (t >= te) t = te - 1; mark(t, 1); }
                if (last) { uint32_t t = tb + (uint32_t)(en >> 6) + 1; if (t > te) t = te; mark(t, -1); }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SPAN_H; i += SPAN_THREADS)
        if (hkey[i] != 0xFFFFFFFFu && hval[i] != 0) atomicAdd(diff + hkey[i], hval[i]);
}

// Upper bound on the reads the reference's pileup engine holds at once while it walks one cell type's BAM
// (bam.pileup(..., max_depth = 200000), BaseCellCounter.py:191): reads of that cell type whose span overlaps a 64-position
// tile, maximum over tiles and cell types.  Evaluated on request (lsg_max_live_reads), cached until reads or barcodes change.
static int live_read_bound_impl(lsg_ctx* c, bool by_ct, int64_t* out);

int live_read_bound(lsg_ctx* c) {
    if (c->max_live_reads >= 0) return 0;
    return live_read_bound_impl(c, c->n_ct > 0 && c->n_cb > 0, &c->max_live_reads);
}

// the same bound over every read that carries a barcode, whatever its cell type: does not depend on the barcode table (cached per
// load) and is >= the per-cell-type bound, so "all reads <= max_depth" settles the question for every table
int live_read_bound_all(lsg_ctx* c) {
    if (c->max_live_all >= 0) return 0;
    return live_read_bound_impl(c, false, &c->max_live_all);
}

static int live_read_bound_impl(lsg_ctx* c, bool by_ct, int64_t* out) {
    const int64_t S = c->rd.n_segs;
    if (S <= 0 || c->n_tiles == 0) { *out = 0; return 0; }
    hipStream_t st = c->stream;
    const size_t T = (size_t)c->n_tiles + 1;
    DevBuf diff, run, tmp, mx;
    auto fail = [&](int rc) { diff.release(); run.release(); tmp.release(); mx.release(); return rc; };
    if (diff.reserve(T * 4) || run.reserve(T * 4) || mx.reserve(64)) return fail(-1);
    size_t tb = 0, tb2 = 0;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, tb, diff.as<int32_t>(), run.as<int32_t>(), (int)T, st) != hipSuccess ||
        hipcub::DeviceReduce::Max(nullptr, tb2, run.as<int32_t>(), mx.as<int32_t>(), (int)T, st) != hipSuccess ||
        tmp.reserve((tb > tb2 ? tb : tb2) + 256)) return fail(-1);
    int64_t best = 0;
    for (int ct = 0; ct < (by_ct ? c->n_ct : 1); ++ct) {
        if (hipMemsetAsync(diff.p, 0, T * 4, st) != hipSuccess) { set_error("live bound: memset failed"); return fail(-1); }
        hipLaunchKernelGGL(k_span_marks, dim3((unsigned)((S + SPAN_THREADS - 1) / SPAN_THREADS)), dim3(SPAN_THREADS), 0, st, c->rd.seg_read, c->rd.seg_start, c->rd.seg_len, S,
                           c->rd.read_tid, c->rd.read_cb, by_ct ? c->d_celltype_of.as<uint8_t>() : (const uint8_t*)nullptr, c->n_cb, ct,
                           c->d_tile_base.as<uint32_t>(), c->n_contigs, diff.as<int32_t>());
        tb = tb2 = tmp.cap;
        if (hipcub::DeviceScan::InclusiveSum(tmp.p, tb, diff.as<int32_t>(), run.as<int32_t>(), (int)T, st) != hipSuccess ||
            hipcub::DeviceReduce::Max(tmp.p, tb2, run.as<int32_t>(), mx.as<int32_t>(), (int)T, st) != hipSuccess) { set_error("live bound: scan failed"); return fail(-1); }
        int32_t m = 0;
        if (hipMemcpyAsync(&m, mx.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("live bound: %s", hipGetErrorString(hipGetLastError())); return fail(-1); }
        if (m > best) best = m;
    }
    *out = best;
    return fail(0);
}

// The same question at POSITION resolution, asked only when the tile-level bound cannot say no (deep samples: C4's tiles hold more than
// 200 000 reads that its positions never do, and the host replay of the rule over 50 M reads takes seconds): per cell type, +1 at a read's
// first column and -1 behind the column after its last one (where the buffer lets go of it), summed along every contig; the maximum
// over positions and cell types is the largest number of reads any buffer can hold when a read is pushed, counting that read: a push is
// refused iff (reads buffered) + 1 > max_depth, and (reads buffered) + 1 <= that maximum.  Pools ignore the read filters and the
// pileup windows (both only shrink a pool).  One pass of atomics (the reads of a deep gene pile up on a few cache lines: merged per
// workgroup would be the next step), a scan and a reduction over the genome's positions per cell type: ~10 ms at C4.
__global__ void k_exact_marks(const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, const int32_t* read_tid, const int32_t* read_cb,
                              const uint8_t* celltype_of, int32_t n_cb, int32_t ct, const int64_t* pos_base, const int64_t* contig_len, int32_t n_contigs, int32_t* diff) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    const uint32_t r = seg_read[s];
    const int32_t tid = read_tid[r], cb = read_cb[r];
    if (cb < 0 || cb >= n_cb || tid < 0 || tid >= n_contigs || celltype_of[cb] != ct) return;
    const bool first = s == 0 || seg_read[s - 1] != r, last = s + 1 == n_segs || seg_read[s + 1] != r;
    if (!first && !last) return;
    const int64_t L = contig_len[tid], base = pos_base[tid];              // the contig's slots: positions 0 .. L + 1
    int64_t st = seg_start[s], en = st + (seg_len[s] > 0 ? seg_len[s] : 0);
    st = st < 0 ? 0 : (st > L ? L : st);
    en = en < 0 ? 0 : (en > L ? L : en);
    if (first) atomicAdd(diff + base + st, 1);
    if (last) atomicAdd(diff + base + en + 1, -1);
}

int live_read_bound_exact(lsg_ctx* c) {
    if (c->max_live_exact >= 0) return 0;
    const int64_t S = c->rd.n_segs;
    if (S <= 0 || c->n_ct <= 0 || c->n_cb <= 0) { c->max_live_exact = 0; return 0; }
    hipStream_t st = c->stream;
    std::vector<int64_t> base((size_t)c->n_contigs + 1, 0);
    for (int t = 0; t < c->n_contigs; ++t) base[(size_t)t + 1] = base[(size_t)t] + c->contig_len[(size_t)t] + 2;
    const int64_t P = base[(size_t)c->n_contigs];
    if (P >= (1ll << 31)) {        // (the scan's 32-bit item count: a genome beyond 2^31 positions keeps the tile-level answer - the host replay decides)
        c->max_live_exact = c->max_live_reads >= 0 ? c->max_live_reads : INT64_MAX / 2;
        return 0;
    }
    DevBuf diff, run, tmp, mx, dbase;
    auto fail = [&](int rc) { diff.release(); run.release(); tmp.release(); mx.release(); dbase.release(); return rc; };
    if (diff.reserve((size_t)P * 4 + 64) || run.reserve((size_t)P * 4 + 64) || mx.reserve(64) || dbase.reserve(base.size() * 8)) return fail(-1);
    size_t tb = 0, tb2 = 0;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, tb, diff.as<int32_t>(), run.as<int32_t>(), (int)P, st) != hipSuccess ||
        hipcub::DeviceReduce::Max(nullptr, tb2, run.as<int32_t>(), mx.as<int32_t>(), (int)P, st) != hipSuccess ||
        tmp.reserve((tb > tb2 ? tb : tb2) + 256)) return fail(-1);
    if (hipMemcpyAsync(dbase.p, base.data(), base.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail(-1);
    int64_t best = 0;
    for (int ct = 0; ct < c->n_ct; ++ct) {
        if (hipMemsetAsync(diff.p, 0, (size_t)P * 4, st) != hipSuccess) { set_error("exact live bound: memset failed"); return fail(-1); }
        hipLaunchKernelGGL(k_exact_marks, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, c->rd.seg_read, c->rd.seg_start, c->rd.seg_len, S, c->rd.read_tid, c->rd.read_cb,
                           c->d_celltype_of.as<uint8_t>(), c->n_cb, ct, dbase.as<int64_t>(), c->d_contig_len.as<int64_t>(), c->n_contigs, diff.as<int32_t>());
        tb = tb2 = tmp.cap;
        if (hipcub::DeviceScan::InclusiveSum(tmp.p, tb, diff.as<int32_t>(), run.as<int32_t>(), (int)P, st) != hipSuccess ||
            hipcub::DeviceReduce::Max(tmp.p, tb2, run.as<int32_t>(), mx.as<int32_t>(), (int)P, st) != hipSuccess) { set_error("exact live bound: scan failed"); return fail(-1); }
        int32_t m = 0;
        if (hipMemcpyAsync(&m, mx.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("exact live bound: %s", hipGetErrorString(hipGetLastError())); return fail(-1); }
        if (m > best) best = m;
    }
    c->max_live_exact = best;
    return fail(0);
}

// first reference position after a read's last pileup column (its last segment's end; pos + 1 without segments)
__global__ void k_read_end(const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, int32_t* read_end) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    const uint32_t r = seg_read[s];
    if (s + 1 == n_segs || seg_read[s + 1] != r) read_end[r] = seg_start[s] + seg_len[s];
}
__global__ void k_read_end_init(const int32_t* read_pos, int64_t n_reads, int32_t* read_end) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) read_end[r] = read_pos[r] + 1;
}

// bam.pileup(CHROM, START, END, ..., max_depth) as htslib applies it (sam.c bam_plp_push / bam_plp_next), per cell type's read stream
// AND per window of the reference (BaseCellCounter.py:185-191: a fresh pileup for every [1 + k W, 1 + (k + 1) W), W = --bin 50000).
// Stream of cell type c = the records SplitBamCellTypes wrote to its BAM (barcode of that type, MAPQ >= min_mq) that the pileup's read
// filter lets in (flag_exclude without the supplementary bit, which only BaseCellCounter.py:249 tests later; ignore_orphans; min_mq);
// pool of a window = the stream's reads overlapping it (the index fetch), in coordinate order: the first read of a start position P
// always enters the buffer; every later read starting at P is dropped iff (reads buffered, i.e. entered and ending at or after P) + 1 >
// max_depth — mp->cnt counts the spare tail node, and reads whose last column was P - 1 are only freed while column P is swept.  A read
// that overlaps two windows passes through two buffers and may be dropped by one and counted by the other (the second window's buffer
// never held the reads that ended before it): the verdict is per (read, window), and the store's entries never cross a window edge.
// Sequential by nature (what was dropped decides what is buffered), so it runs on the host — but only when lsg_max_live_reads() says a
// buffer can reach max_depth at all.  Result: d_read_drop[r] = 1 dropped in every window r overlaps, 2 in some of them, listed as
// (r << 32 | window of its contig) in d_drop_pairs (sorted); pileup.hip k_read_stats / k_tm_resolve apply them.
int depth_cap_drops(lsg_ctx* c, const lsg_count_params* p) {
    c->has_drops = false; c->n_depth_dropped = 0; c->n_drop_pairs = 0;
    if (p->max_depth <= 0 || c->rd.n_reads <= 0 || c->n_ct <= 0) return 0;
    if (live_read_bound_all(c)) return -1;
    if (c->max_live_all + 1 <= (int64_t)p->max_depth) return 0;           // not even all reads together fill a buffer: nothing is ever dropped
    if (live_read_bound(c)) return -1;
    if (c->max_live_reads + 1 <= (int64_t)p->max_depth) return 0;          // no cell type's buffer can exceed the cap
    if (live_read_bound_exact(c)) return -1;
    if (c->max_live_exact <= (int64_t)p->max_depth) return 0;              // ... not at any POSITION (the tiles over-count): a push is refused iff buffered + 1 > max_depth
    hipStream_t st = c->stream;
    const int64_t R = c->rd.n_reads, S = c->rd.n_segs, W = c->st_window;
    DevBuf d_end;
    if (d_end.reserve((size_t)R * 4)) return -1;
    hipLaunchKernelGGL(k_read_end_init, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, c->rd.read_pos, R, d_end.as<int32_t>());
    if (S > 0) hipLaunchKernelGGL(k_read_end, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, c->rd.seg_read, c->rd.seg_start, c->rd.seg_len, S, d_end.as<int32_t>());
    std::vector<int32_t> tid((size_t)R), pos((size_t)R), end((size_t)R), cb((size_t)R); std::vector<uint16_t> flag((size_t)R); std::vector<uint8_t> mapq((size_t)R);
    std::vector<uint8_t> ctof((size_t)c->n_cb);
    auto cp = [&](void* dst, const void* src, size_t n) { return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st) != hipSuccess; };
    if (cp(tid.data(), c->rd.read_tid, (size_t)R * 4) || cp(pos.data(), c->rd.read_pos, (size_t)R * 4) || cp(end.data(), d_end.p, (size_t)R * 4) ||
        cp(cb.data(), c->rd.read_cb, (size_t)R * 4) || cp(flag.data(), c->rd.read_flag, (size_t)R * 2) || cp(mapq.data(), c->rd.read_mapq, (size_t)R) ||
        cp(ctof.data(), c->d_celltype_of.p, (size_t)c->n_cb) || hipStreamSynchronize(st) != hipSuccess) { d_end.release(); set_error("depth cap: copy failed"); return -1; }
    d_end.release();
    // coordinate order (a decoded BAM already is; the synthetic generator's read index is gene order)
    std::vector<uint32_t> order((size_t)R);
    for (int64_t i = 0; i < R; ++i) order[(size_t)i] = (uint32_t)i;
    auto key = [&](uint32_t i) { return ((uint64_t)(uint32_t)tid[i] << 32) | (uint32_t)pos[i]; };
    bool sorted = true;
    for (int64_t i = 1; i < R && sorted; ++i) sorted = key((uint32_t)(i - 1)) <= key((uint32_t)i);
    if (!sorted) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
    std::vector<uint16_t> n_over((size_t)R, 0), n_drop((size_t)R, 0);      // windows a read overlaps / is dropped in (a read reaches over a handful at most)
    std::vector<uint64_t> pairs;
    const uint32_t pool_flags = p->flag_exclude & ~0x800u;
    std::vector<uint32_t> stream, pool, carry;
    std::vector<int32_t> heap;                                             // min-heap of the buffered reads' ends
    auto cmp = [](int32_t a, int32_t b) { return a > b; };
    for (int ct = 0; ct < c->n_ct; ++ct) {
        int64_t k0 = 0;
        while (k0 < R) {                                                   // one contig of the coordinate order at a time
            const int32_t t = tid[order[(size_t)k0]];
            int64_t k1 = k0;
            stream.clear();
            for (; k1 < R && tid[order[(size_t)k1]] == t; ++k1) {
                const uint32_t i = order[(size_t)k1];
                if (t < 0 || t >= c->n_contigs || cb[i] < 0 || cb[i] >= c->n_cb || ctof[(size_t)cb[i]] != ct) continue;
                if ((int)mapq[i] < p->min_mq || (flag[i] & pool_flags)) continue;
                if (p->ignore_orphans && (flag[i] & 1u) && !(flag[i] & 2u)) continue;
                stream.push_back(i);
            }
            k0 = k1;
            size_t nxt = 0;
            carry.clear();
            int64_t w = 0;
            while (nxt < stream.size() || !carry.empty()) {
                if (carry.empty()) { const int64_t p0 = pos[stream[nxt]]; const int64_t wn = p0 >= 1 ? (p0 - 1) / W : 0; if (wn > w) w = wn; }      // nothing reaches into the windows in between
                const int64_t ws = 1 + W * w, we = ws + W;
                pool.assign(carry.begin(), carry.end());                  // (reads of earlier windows that reach into this one, in order) + the reads starting below its end
                while (nxt < stream.size() && (int64_t)pos[stream[nxt]] < we) { if ((int64_t)end[stream[nxt]] > ws) pool.push_back(stream[nxt]); ++nxt; }
                heap.clear();
                int32_t cur_pos = -1; bool first_here = true;
                for (uint32_t i : pool) {
                    ++n_over[i];
                    if (pos[i] != cur_pos) {
                        cur_pos = pos[i]; first_here = true;
                        while (!heap.empty() && heap.front() < cur_pos) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.pop_back(); }      // ended before P: freed
                    }
                    if (!first_here && (int64_t)heap.size() + 1 > (int64_t)p->max_depth) { ++n_drop[i]; pairs.push_back(((uint64_t)i << 32) | (uint64_t)w); continue; }
                    first_here = false;
                    heap.push_back(end[i]); std::push_heap(heap.begin(), heap.end(), cmp);
                }
                carry.clear();
                for (uint32_t i : pool) if ((int64_t)end[i] > we) carry.push_back(i);
                ++w;
            }
        }
    }
    if (pairs.empty()) return 0;
    std::vector<uint8_t> drop((size_t)R, 0);
    int64_t n_dropped = 0;
    for (int64_t i = 0; i < R; ++i) if (n_drop[(size_t)i]) { drop[(size_t)i] = n_drop[(size_t)i] == n_over[(size_t)i] ? 1 : 2; ++n_dropped; }
    std::vector<uint64_t> some;
    for (uint64_t pr : pairs) if (drop[(size_t)(pr >> 32)] == 2) some.push_back(pr);
    std::sort(some.begin(), some.end());
    if (c->d_read_drop.reserve((size_t)R) || c->d_drop_pairs.reserve((some.size() + 1) * 8)) return -1;
    if (hipMemcpyAsync(c->d_read_drop.p, drop.data(), (size_t)R, hipMemcpyHostToDevice, st) != hipSuccess ||
        (!some.empty() && hipMemcpyAsync(c->d_drop_pairs.p, some.data(), some.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess) { set_error("depth cap: upload failed"); return -1; }
    c->has_drops = true; c->n_depth_dropped = n_dropped; c->n_drop_pairs = (int64_t)some.size();
    return 0;
}

} // namespace lsg
