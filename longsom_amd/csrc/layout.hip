// Event layout in HBM.  The C-ABI takes the events of a segment at any offset (seg_ev_off); on load the library
// re-lays them out TILE-ALIGNED in its own buffer: a segment starting at reference position p gets whole 64-position
// tiles, and the event of position q sits at  tile_slot * 64 + (q & 63).  A pileup entry (= one segment x one 64-position
// tile) then lies inside exactly one aligned 128-byte line, so the walk kernels fetch one line per entry instead of the
// ~1.7 lines an arbitrary 2-byte alignment costs (measured: 31.5 GB -> see DESIGN.md), and neighbouring tiles of a
// segment never share a line.  Padding events are 0 (= not countable) and are never read.
#include "lsg_ctx.h"
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
    int64_t st = seg_start[s], en = st + (seg_len[s] > 0 ? seg_len[s] - 1 : 0);
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
int live_read_bound(lsg_ctx* c) {
    if (c->max_live_reads >= 0) return 0;
    const int64_t S = c->rd.n_segs;
    if (S <= 0 || c->n_tiles == 0) { c->max_live_reads = 0; return 0; }
    hipStream_t st = c->stream;
    const size_t T = (size_t)c->n_tiles + 1;
    DevBuf diff, run, tmp, mx;
    auto fail = [&](int rc) { diff.release(); run.release(); tmp.release(); mx.release(); return rc; };
    if (diff.reserve(T * 4) || run.reserve(T * 4) || mx.reserve(64)) return fail(-1);
    size_t tb = 0, tb2 = 0;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, tb, diff.as<int32_t>(), run.as<int32_t>(), (int)T, st) != hipSuccess ||
        hipcub::DeviceReduce::Max(nullptr, tb2, run.as<int32_t>(), mx.as<int32_t>(), (int)T, st) != hipSuccess ||
        tmp.reserve((tb > tb2 ? tb : tb2) + 256)) return fail(-1);
    const bool by_ct = c->n_ct > 0 && c->n_cb > 0;
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
    c->max_live_reads = best;
    return fail(0);
}

int relayout_events(lsg_ctx* c) {
    const int64_t S = c->rd.n_segs;
    c->max_live_reads = -1;
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
