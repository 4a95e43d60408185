// Event layout in HBM.  The C-ABI takes the events of a segment at any offset (seg_ev_off); on load the library
// re-lays them out TILE-ALIGNED in its own buffer: a segment starting at reference position p gets whole 64-position
// tiles, and the event of position q sits at  tile_slot * 64 + (q & 63).  A pileup entry (= one segment x one 64-position
// tile) then lies inside exactly one aligned 128-byte line, so the walk kernels fetch one line per entry instead of the
// ~1.7 lines an arbitrary 2-byte alignment costs (measured: 31.5 GB -> see DESIGN.md), and neighbouring tiles of a
// segment never share a line.  Padding events are 0 (= not countable) and are never read.
#include "lsg_ctx.h"
#include <algorithm>
#include <vector>
#include <hipcub/hipcub.hpp>

namespace lsg {

__global__ void k_seg_slots(const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, int64_t* slot_events) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_segs) return;
    int64_t v = 0;
    if (s < n_segs) {
        const int64_t st = seg_start[s], ln = seg_len[s];
        if (ln > 0) v = (((st & 63) + ln + 63) >> 6) << 6;
    }
    slot_events[s] = v;
}

// one wavefront per segment: coalesced copy of its events to the aligned slot
__global__ __launch_bounds__(256) void k_relayout(const uint16_t* src, int64_t n_src, const int32_t* seg_start, const int32_t* seg_len,
                                                  const int64_t* old_off, const int64_t* slot_base, int64_t n_segs,
                                                  const uint32_t* seg_read, int64_t n_reads,
                                                  uint16_t* dst, int64_t* new_off, uint32_t* bad) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t s = wave; s < n_segs; s += n_waves) {
        const int64_t o = old_off[s], ln = seg_len[s], st = seg_start[s];
        const int64_t d = slot_base[s] + (st & 63);
        if (lane == 0) { new_off[s] = d; if ((int64_t)seg_read[s] >= n_reads) atomicOr(bad, 2u); }
        if (ln <= 0) continue;
        if (o < 0 || o + ln > n_src) { if (lane == 0) atomicOr(bad, 1u); continue; }
        for (int64_t i = lane; i < ln; i += 64) dst[d + i] = src[o + i];
    }
}

// +1 at the tile a read's first segment starts in, -1 past the tile its last segment ends in; reads of cell type ct
// (celltype_of == nullptr: every read with a known barcode)
__global__ void k_span_marks(const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs,
                             const int32_t* read_tid, const int32_t* read_cb, const uint8_t* celltype_of, int32_t n_cb, int32_t ct,
                             const uint32_t* tile_base, int32_t n_contigs, int32_t* diff) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    const uint32_t r = seg_read[s];
    const int32_t tid = read_tid[r];
    const int32_t cb = read_cb[r];
    if (cb < 0 || tid < 0 || tid >= n_contigs) return;
    if (celltype_of && (cb >= n_cb || celltype_of[cb] != ct)) return;
    const bool first = s == 0 || seg_read[s - 1] != r, last = s + 1 == n_segs || seg_read[s + 1] != r;
    if (!first && !last) return;
    // the read is still buffered while the column AFTER its last one is entered (freed by that column's sweep): span end inclusive
    int64_t st = seg_start[s], en = st + (seg_len[s] > 0 ? seg_len[s] : 0);
    const uint32_t tb = tile_base[tid], te = tile_base[tid + 1];
    if (te <= tb) return;
    if (st < 0) st = 0;
    if (en < 0) en = 0;
    // both marks are clamped into the contig so that every +1 has its -1 (a bound must never under-count)
    if (first) { uint32_t t = tb + (uint32_t)(st >> 6); if (t >= te) t = te - 1; atomicAdd(diff + t, 1); }
    if (last) { uint32_t t = tb + (uint32_t)(en >> 6) + 1; if (t > te) t = te; atomicSub(diff + t, 1); }
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
        hipLaunchKernelGGL(k_span_marks, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, c->rd.seg_read, c->rd.seg_start, c->rd.seg_len, S,
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

// bam.pileup(..., max_depth) as htslib applies it (sam.c bam_plp_push / bam_plp_next; BaseCellCounter.py:191), per cell type's read
// stream.  Stream of cell type c = the records SplitBamCellTypes wrote to its BAM (barcode of that type, MAPQ >= min_mq) that the
// pileup's read filter lets in (flag_exclude without the supplementary bit, which only BaseCellCounter.py:249 tests later;
// ignore_orphans; min_mq).  In coordinate order: the first read of a start position P always enters the buffer; every later read
// starting at P is dropped iff (reads buffered, i.e. entered and ending at or after P) + 1 > max_depth — mp->cnt counts the spare
// tail node, and reads whose last column was P - 1 are only freed while column P is swept.  Sequential by nature (what was dropped
// decides what is buffered), so it runs on the host — but only when lsg_max_live_reads() says a buffer can reach max_depth at all.
int depth_cap_drops(lsg_ctx* c, const lsg_count_params* p) {
    c->has_drops = false; c->n_depth_dropped = 0;
    if (p->max_depth <= 0 || c->rd.n_reads <= 0 || c->n_ct <= 0) return 0;
    if (live_read_bound_all(c)) return -1;
    if (c->max_live_all + 1 <= (int64_t)p->max_depth) return 0;           // not even all reads together fill a buffer: nothing is ever dropped
    if (live_read_bound(c)) return -1;
    if (c->max_live_reads + 1 <= (int64_t)p->max_depth) return 0;          // no cell type's buffer can exceed the cap
    hipStream_t st = c->stream;
    const int64_t R = c->rd.n_reads, S = c->rd.n_segs;
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
    std::vector<uint8_t> drop((size_t)R, 0);
    const uint32_t pool_flags = p->flag_exclude & ~0x800u;
    int64_t n_drop = 0;
    for (int ct = 0; ct < c->n_ct; ++ct) {
        std::vector<int32_t> heap;                                         // min-heap of the buffered reads' ends
        auto cmp = [](int32_t a, int32_t b) { return a > b; };
        int32_t cur_tid = -1, cur_pos = -1; bool first_here = true;
        for (int64_t k = 0; k < R; ++k) {
            const uint32_t i = order[(size_t)k];
            if (tid[i] < 0 || tid[i] >= c->n_contigs || cb[i] < 0 || cb[i] >= c->n_cb || ctof[(size_t)cb[i]] != ct) continue;
            if ((int)mapq[i] < p->min_mq || (flag[i] & pool_flags)) continue;
            if (p->ignore_orphans && (flag[i] & 1u) && !(flag[i] & 2u)) continue;
            if (tid[i] != cur_tid) { heap.clear(); cur_tid = tid[i]; cur_pos = -1; }
            if (pos[i] != cur_pos) {
                cur_pos = pos[i]; first_here = true;
                while (!heap.empty() && heap.front() < cur_pos) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.pop_back(); }      // ended before P: freed
            }
            if (!first_here && (int64_t)heap.size() + 1 > (int64_t)p->max_depth) { drop[i] = 1; ++n_drop; continue; }
            first_here = false;
            heap.push_back(end[i]); std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    if (n_drop == 0) return 0;
    if (c->d_read_drop.reserve((size_t)R)) return -1;
    if (hipMemcpyAsync(c->d_read_drop.p, drop.data(), (size_t)R, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("depth cap: upload failed"); return -1; }
    c->has_drops = true; c->n_depth_dropped = n_drop;
    return 0;
}

int relayout_events(lsg_ctx* c) {
    const int64_t S = c->rd.n_segs;
    c->max_live_reads = -1; c->max_live_all = -1;
    if (S <= 0) { c->rd.n_events = 0; return 0; }
    hipStream_t st = c->stream;
    DevBuf slots, base, noff, tmp, flag, aligned;
    auto fail = [&](int rc) { slots.release(); base.release(); noff.release(); tmp.release(); flag.release(); aligned.release(); return rc; };
    if (slots.reserve((size_t)(S + 1) * 8) || base.reserve((size_t)(S + 1) * 8) || noff.reserve((size_t)(S + 1) * 8) || flag.reserve(64)) return fail(-1);
    hipLaunchKernelGGL(k_seg_slots, dim3((unsigned)((S + 1 + 255) / 256)), dim3(256), 0, st, c->rd.seg_start, c->rd.seg_len, S, slots.as<int64_t>());
    size_t tb = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tb, slots.as<int64_t>(), base.as<int64_t>(), (int)(S + 1), st) != hipSuccess || tmp.reserve(tb + 256)) return fail(-1);
    tb = tmp.cap;
    if (hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, slots.as<int64_t>(), base.as<int64_t>(), (int)(S + 1), st) != hipSuccess) { set_error("relayout: scan failed"); return fail(-1); }
    int64_t E2 = 0;
    if (hipMemcpyAsync(&E2, base.as<int64_t>() + S, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("relayout: copy failed"); return fail(-1); }
    if (E2 >= (1ll << 40)) { set_error("lsg_load_reads: more than 2^40 events after tile alignment"); return fail(-2); }
    if (aligned.reserve((size_t)E2 * 2 + 256)) return fail(-1);
    if (hipMemsetAsync(aligned.p, 0, (size_t)E2 * 2 + 256, st) != hipSuccess || hipMemsetAsync(flag.p, 0, 64, st) != hipSuccess) { set_error("relayout: memset failed"); return fail(-1); }
    unsigned grid = (unsigned)((S + 3) / 4 < (int64_t)c->n_cus * 32 ? (S + 3) / 4 : (int64_t)c->n_cus * 32);
    hipLaunchKernelGGL(k_relayout, dim3(grid), dim3(256), 0, st, c->rd.events, c->rd.n_events, c->rd.seg_start, c->rd.seg_len, c->rd.seg_ev_off,
                       base.as<int64_t>(), S, c->rd.seg_read, c->rd.n_reads, aligned.as<uint16_t>(), noff.as<int64_t>(), flag.as<uint32_t>());
    uint32_t bad = 0;
    if (hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("relayout: kernel failed: %s", hipGetErrorString(hipGetLastError())); return fail(-1); }
    if (bad & 2u) { set_error("lsg_load_reads: a segment's read index lies outside the read arrays"); return fail(-2); }
    if (bad) { set_error("lsg_load_reads: a segment's event range lies outside the events array"); return fail(-2); }
    // the library's own copies replace whatever the caller handed over
    if (c->b_seg_ev_off.reserve((size_t)(S + 1) * 8)) return fail(-1);
    if (hipMemcpyAsync(c->b_seg_ev_off.p, noff.p, (size_t)S * 8, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("relayout: copy failed"); return fail(-1); }
    c->b_events.release();
    c->b_events = aligned; aligned.p = nullptr; aligned.cap = 0;
    c->rd.events = c->b_events.as<uint16_t>();
    c->rd.seg_ev_off = c->b_seg_ev_off.as<int64_t>();
    c->rd.n_events = E2;
    return fail(0);
}

} // namespace lsg
