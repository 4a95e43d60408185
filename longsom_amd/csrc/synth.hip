// Device-side generator of the synthetic workload (synth_model.h) straight into HBM, so that the
// bench's 10M-read / 1.2e10-event configuration never crosses PCIe.  Not on the timed path.
#include "lsg_ctx.h"
#include "synth_model.h"
#include <hipcub/hipcub.hpp>

namespace lsg {

__global__ void k_synth_header(lsg_synth_model m, int32_t* read_tid, int32_t* read_pos, uint16_t* read_flag, uint8_t* read_mapq,
                               int32_t* read_cb, int64_t* ev_cnt, uint32_t* seg_cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > m.n_reads) return;
    if (i == m.n_reads) { ev_cnt[i] = 0; seg_cnt[i] = 0; return; }
    sm_read r;
    sm_read_header(&m, i + m.read_base, &r);
    read_tid[i] = r.tid;
    read_pos[i] = m.exon_start[r.e0] + (r.t_off - m.exon_cum[r.e0]);
    read_flag[i] = r.flag;
    read_mapq[i] = r.mapq;
    read_cb[i] = r.cb >= 0 ? r.cb : -1;
    ev_cnt[i] = sm_read_region(&m, &r);
    seg_cnt[i] = (uint32_t)(r.e1 - r.e0 + 1);
}

// one workgroup per read: threads stride over the read's transcript coordinates
__global__ __launch_bounds__(256) void k_synth_fill(lsg_synth_model m, const int64_t* ev_off, const uint32_t* seg_off,
                                                    uint32_t* seg_read, int32_t* seg_start, int32_t* seg_len, int64_t* seg_ev_off,
                                                    uint16_t* events) {
    for (int64_t i = blockIdx.x; i < m.n_reads; i += gridDim.x) {
        sm_read r;
        const int64_t ig = i + m.read_base;
        sm_read_header(&m, ig, &r);
        const int64_t eo = ev_off[i];
        const uint32_t so = seg_off[i];
        const int32_t end = r.t_off + r.t_len;
        const bool phased = m.layout == LSG_LAYOUT_PHASED;
        if ((int)threadIdx.x <= r.e1 - r.e0) {
            int32_t st = 0, ln = 0; int64_t cur = 0, at = 0;
            for (int32_t k = 0; k <= (int)threadIdx.x; ++k) { sm_read_segment(&m, &r, k, &st, &ln); at = phased ? sm_phase_place(cur, st) : cur; cur = at + ln; }
            seg_read[so + threadIdx.x] = (uint32_t)i;
            seg_start[so + threadIdx.x] = st;
            seg_len[so + threadIdx.x] = ln;
            seg_ev_off[so + threadIdx.x] = eo + at;
        }
        // (the events: a thread walks the exons its transcript coordinates fall into, and with them the segments' places)
        int32_t x = r.e0, st = 0, ln = 0; int64_t at = 0;
        sm_read_segment(&m, &r, 0, &st, &ln);
        if (phased) at = sm_phase_place(0, st);
        int32_t seg_t0 = r.t_off;                                   // transcript coordinate of the segment's first event
        for (int32_t j = r.t_off + (int)threadIdx.x; j < end; j += (int)blockDim.x) {
            while (m.exon_cum[x] + m.exon_len[x] <= j) {
                ++x;
                const int64_t cur = at + ln;
                seg_t0 += ln;
                sm_read_segment(&m, &r, x - r.e0, &st, &ln);
                at = phased ? sm_phase_place(cur, st) : cur;
            }
            const int32_t xt0 = m.exon_cum[x];
            events[eo + at + (j - seg_t0)] = sm_event(&m, ig, &r, j, xt0, xt0 + m.exon_len[x], m.exon_start[x]);
        }
    }
}

__global__ void k_synth_ref(uint64_t seed, int32_t tid, int64_t len, uint8_t* out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < len) out[p] = sm_ref_base(seed, tid, p);
}

template <class T> static int up(lsg_ctx* c, DevBuf& b, const T* src, size_t n) {
    if (b.reserve(n * sizeof(T) + 16)) return -1;
    LSG_HIP(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return 0;
}
} // namespace lsg

using namespace lsg;

extern "C" {

int lsg_synth_reference(lsg_ctx* c, uint64_t seed) {
    if (!c || c->n_contigs <= 0) { set_error("lsg_synth_reference: set contigs first"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    for (int t = 0; t < c->n_contigs; ++t) {
        int64_t len = c->contig_len[t];
        if (c->ref[t].reserve((size_t)(len > 0 ? len : 1))) return -1;
        if (len > 0)
            hipLaunchKernelGGL(k_synth_ref, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, seed, t, len, c->ref[t].as<uint8_t>());
        c->ref_ptr[t] = c->ref[t].as<uint8_t>();
    }
    LSG_HIP(hipMemcpyAsync(c->d_ref_ptrs.p, c->ref_ptr.data(), (size_t)c->n_contigs * sizeof(void*), hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    c->counted = c->called = false;
    return 0;
}

// model: host pointers (gene tables are copied); celltype_of is taken from lsg_set_barcodes.
// Generates the model's compact read-record arrays into buffers of the handle and describes them in *out (device pointers, valid
// until the next generate / lsg_synth_reads / lsg_destroy): what a caller with device-resident arrays hands to lsg_load_reads.
int lsg_synth_generate(lsg_ctx* c, const lsg_synth_model* hm, lsg_reads* out) {
    if (!c || !hm || !out) { set_error("lsg_synth_generate: bad arguments"); return -2; }
    if (c->n_cb <= 0 || hm->n_cb != c->n_cb) { set_error("lsg_synth_generate: set barcodes first (n_cb %d vs %d)", hm->n_cb, c->n_cb); return -2; }
    if (hm->n_genes <= 0 || hm->n_reads < 0) { set_error("lsg_synth_generate: empty model"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int G = hm->n_genes;
    const int X = hm->gene_exon_off[G];
    const int64_t R = hm->n_reads;
    DevBuf &g_tid = c->syn[0], &g_xoff = c->syn[1], &x_start = c->syn[2], &x_len = c->syn[3], &x_cum = c->syn[4], &g_roff = c->syn[5],
           &evcnt = c->syn[6], &segcnt = c->syn[7], &evoff = c->syn[8], &segoff = c->syn[9], &tmpb = c->syn[10];
    DevBuf &o_tid = c->gen[0], &o_pos = c->gen[1], &o_flag = c->gen[2], &o_mapq = c->gen[3], &o_cb = c->gen[4], &o_sread = c->gen[5], &o_sstart = c->gen[6],
           &o_slen = c->gen[7], &o_sevoff = c->gen[8], &o_events = c->gen[9];
    if (up(c, g_tid, hm->gene_tid, G) || up(c, g_xoff, hm->gene_exon_off, G + 1) || up(c, x_start, hm->exon_start, X) ||
        up(c, x_len, hm->exon_len, X) || up(c, x_cum, hm->exon_cum, X) || up(c, g_roff, hm->gene_read_off, G + 1)) return -1;
    lsg_synth_model m = *hm;
    m.gene_tid = g_tid.as<int32_t>(); m.gene_exon_off = g_xoff.as<int32_t>(); m.exon_start = x_start.as<int32_t>();
    m.exon_len = x_len.as<int32_t>(); m.exon_cum = x_cum.as<int32_t>(); m.gene_read_off = g_roff.as<int64_t>();
    m.celltype_of = c->d_celltype_of.as<uint8_t>();

    if (o_tid.reserve((R + 1) * 4) || o_pos.reserve((R + 1) * 4) || o_flag.reserve((R + 1) * 2) ||
        o_mapq.reserve(R + 1) || o_cb.reserve((R + 1) * 4) || evcnt.reserve((R + 1) * 8) || segcnt.reserve((R + 1) * 4) ||
        evoff.reserve((R + 1) * 8) || segoff.reserve((R + 1) * 4)) return -1;
    hipLaunchKernelGGL(k_synth_header, dim3((unsigned)((R + 1 + 255) / 256)), dim3(256), 0, st, m, o_tid.as<int32_t>(),
                       o_pos.as<int32_t>(), o_flag.as<uint16_t>(), o_mapq.as<uint8_t>(), o_cb.as<int32_t>(),
                       evcnt.as<int64_t>(), segcnt.as<uint32_t>());
    size_t t1 = 0, t2 = 0;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t1, evcnt.as<int64_t>(), evoff.as<int64_t>(), (int)(R + 1), st));
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, segcnt.as<uint32_t>(), segoff.as<uint32_t>(), (int)(R + 1), st));
    if (tmpb.reserve((t1 > t2 ? t1 : t2) + 16)) return -1;
    size_t t = tmpb.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(tmpb.p, t, evcnt.as<int64_t>(), evoff.as<int64_t>(), (int)(R + 1), st));
    t = tmpb.cap;
    LSG_HIP(hipcub::DeviceScan::ExclusiveSum(tmpb.p, t, segcnt.as<uint32_t>(), segoff.as<uint32_t>(), (int)(R + 1), st));
    int64_t E = 0; uint32_t S = 0;
    LSG_HIP(hipMemcpyAsync(&E, evoff.as<int64_t>() + R, 8, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipMemcpyAsync(&S, segoff.as<uint32_t>() + R, 4, hipMemcpyDeviceToHost, st));
    LSG_HIP(hipStreamSynchronize(st));
    if (o_sread.reserve(((size_t)S + 1) * 4) || o_sstart.reserve(((size_t)S + 1) * 4) || o_slen.reserve(((size_t)S + 1) * 4) ||
        o_sevoff.reserve(((size_t)S + 1) * 8) || o_events.reserve(((size_t)E + 1) * 2)) return -1;
    if (m.layout == LSG_LAYOUT_PHASED && E > 0) LSG_HIP(hipMemsetAsync(o_events.p, 0, (size_t)E * 2, st));      // (the gaps hold 0)
    if (R > 0) {
        unsigned grid = (unsigned)(R < 65536 * 16 ? R : 65536 * 16);
        hipLaunchKernelGGL(k_synth_fill, dim3(grid), dim3(256), 0, st, m, evoff.as<int64_t>(), segoff.as<uint32_t>(),
                           o_sread.as<uint32_t>(), o_sstart.as<int32_t>(), o_slen.as<int32_t>(),
                           o_sevoff.as<int64_t>(), o_events.as<uint16_t>());
    }
    LSG_HIP(hipGetLastError());
    LSG_HIP(hipStreamSynchronize(st));
    *out = lsg_reads{};
    out->n_reads = R; out->n_segs = S; out->n_events = E; out->on_device = 1;
    out->read_tid = o_tid.as<int32_t>(); out->read_pos = o_pos.as<int32_t>();
    out->read_flag = o_flag.as<uint16_t>(); out->read_mapq = o_mapq.as<uint8_t>();
    out->read_cb = o_cb.as<int32_t>(); out->seg_read = o_sread.as<uint32_t>();
    out->seg_start = o_sstart.as<int32_t>(); out->seg_len = o_slen.as<int32_t>();
    out->seg_ev_off = o_sevoff.as<int64_t>(); out->events = o_events.as<uint16_t>();
    c->hint_phased_events = m.layout == LSG_LAYOUT_PHASED ? o_events.p : nullptr;      // (what a load of THESE arrays may take for granted, and checks: build_store)
    return 0;
}

// Generates the model's reads in HBM and loads them (lsg_load_reads of device arrays); the generated arrays are given back.
int lsg_synth_reads(lsg_ctx* c, const lsg_synth_model* hm) {
    lsg_reads g{};
    if (int rc = lsg_synth_generate(c, hm, &g)) return rc;
    const int rc = lsg_load_reads(c, &g);
    for (auto& b : c->gen) b.release();
    return rc;
}

int lsg_get_reads_shape(lsg_ctx* c, int64_t* n_reads, int64_t* n_segs, int64_t* n_events) {
    if (!c) { set_error("lsg_get_reads_shape: NULL handle"); return -2; }
    if (n_reads) *n_reads = c->rd.n_reads;
    if (n_segs) *n_segs = c->rd.n_segs;
    if (n_events) *n_events = c->rd.n_events;
    return 0;
}

// Copies the resident read-record arrays into caller-allocated host arrays (tests / sampling for the CPU baseline).
int lsg_copy_reads_to_host(lsg_ctx* c, const lsg_reads* out) {
    if (!c || !out) { set_error("lsg_copy_reads_to_host: bad arguments"); return -2; }
    if (out->n_reads != c->rd.n_reads || out->n_segs != c->rd.n_segs || out->n_events != c->rd.n_events) {
        set_error("lsg_copy_reads_to_host: shape mismatch"); return -2;
    }
    const bool want_events = out->events != nullptr || out->seg_ev_off != nullptr;      // (both NULL: the per-read and per-segment arrays only — they are always resident)
    if (want_events && c->rd.n_events > 0 && !c->rd.events) { set_error("lsg_copy_reads_to_host: the events were not kept beside the store (lsg_set_keep_reads before the load)"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int64_t R = c->rd.n_reads, S = c->rd.n_segs, E = c->rd.n_events;
#define CP(field, n, sz) if ((n) > 0) LSG_HIP(hipMemcpyAsync((void*)out->field, c->rd.field, (size_t)(n) * (sz), hipMemcpyDeviceToHost, st))
    CP(read_tid, R, 4); if (out->read_pos && c->rd.read_pos) CP(read_pos, R, 4);
    CP(read_flag, R, 2); CP(read_mapq, R, 1); CP(read_cb, R, 4);
    CP(seg_read, S, 4); CP(seg_start, S, 4); CP(seg_len, S, 4);
    if (want_events) { CP(seg_ev_off, S, 8); CP(events, E, 2); }
#undef CP
    LSG_HIP(hipStreamSynchronize(st));
    return 0;
}

int lsg_copy_reference_to_host(lsg_ctx* c, int32_t tid, uint8_t* out) {
    if (!c || !out || tid < 0 || tid >= c->n_contigs || !c->ref_ptr[tid]) { set_error("lsg_copy_reference_to_host: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    if (c->contig_len[tid] > 0) LSG_HIP(hipMemcpy(out, c->ref_ptr[tid], (size_t)c->contig_len[tid], hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
