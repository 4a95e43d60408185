// Per-cell genotyping at target sites: HCCVSingleCellGenotype.py:82-220 and its twin
// SNVCalling/SingleCellGenotype.py:84-228.  One pass over the resident segments: a segment looks its reference range
// up in the sorted target-site keys (binary search over an L2-resident array) and adds the event it carries at every
// target site to that site's per-barcode (Dp, Alt) pair.  No tiles, units or sorting: target sites are a few thousand,
// the pass is bounded by streaming 20 bytes per segment.
#include "lsg_ctx.h"

namespace lsg {

struct GenoArgs {
    int64_t n_reads, n_segs;
    const int32_t* read_tid; const uint16_t* read_flag; const uint8_t* read_mapq; const int32_t* read_cb;
    const uint32_t* seg_read; const int32_t* seg_start; const int32_t* seg_len; const int64_t* seg_ev_off;
    const uint16_t* events;
    const uint8_t* celltype_of; const int64_t* contig_len;
    int32_t n_contigs, n_cb;
    lsg_genotype_params p;
    int64_t n_sites; const int64_t* site_keys; const uint8_t* alt_sym;
    uint32_t* read_cbk;       // admitted barcode per read or 0xFFFFFFFF
    uint32_t* dp; uint32_t* alt;
};

// read admission: pileup flag filter + ignore_orphans + min_mq (HCCVSingleCellGenotype.py:123), not secondary /
// duplicate / supplementary (:168), CB present and in barcodes.tsv (:160-164)
__global__ void k_geno_read_key(GenoArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t flag = a.read_flag[r];
        const int32_t cb = a.read_cb[r], tid = a.read_tid[r];
        bool ok = (flag & a.p.flag_exclude) == 0 && (int)a.read_mapq[r] >= a.p.min_mq && cb >= 0 && cb < a.n_cb && tid >= 0 && tid < a.n_contigs;
        if (ok && a.p.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) ok = false;
        if (ok && a.p.strict_cb && (flag & LSG_FLAG_CB_SUFFIX)) ok = false;
        if (ok && a.celltype_of[cb] == 255) ok = false;
        a.read_cbk[r] = ok ? (uint32_t)cb : 0xFFFFFFFFu;
    }
}

__global__ void k_geno_segments(GenoArgs a) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.n_segs; s += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = a.seg_read[s];
        const uint32_t cb = a.read_cbk[r];
        if (cb == 0xFFFFFFFFu) continue;
        const int32_t tid = a.read_tid[r];
        const int64_t st = a.seg_start[s], ln = a.seg_len[s];
        if (st < 0 || ln <= 0 || st + ln > a.contig_len[tid]) continue;          // malformed: never counted
        const int64_t k_lo = ((int64_t)tid << 32) | st, k_hi = k_lo + ln;
        int64_t lo = 0, hi = a.n_sites;                                           // first site key >= k_lo
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.site_keys[mid] < k_lo) lo = mid + 1; else hi = mid; }
        const int64_t eo = a.seg_ev_off[s];
        for (int64_t i = lo; i < a.n_sites; ++i) {
            const int64_t key = a.site_keys[i];
            if (key >= k_hi) break;
            const uint32_t ev = a.events[eo + (key - k_lo)];
            if (!(ev & LSG_EVENT_VALID) || (int)(ev & 0xffu) < a.p.min_bq) continue;
            const uint32_t sym = (ev >> 8) & 7u;
            if (sym > (uint32_t)LSG_SYM_N) continue;                               // 'O' is not in Bases (:148)
            const bool is_alt = sym == (uint32_t)a.alt_sym[i];
            if (a.p.alt_only && !is_alt) continue;
            const uint64_t cell = (uint64_t)i * (uint64_t)a.n_cb + cb;
            atomicAdd(&a.dp[cell], 1u);
            if (is_alt) atomicAdd(&a.alt[cell], 1u);
        }
    }
}

int run_genotype(lsg_ctx* c, const lsg_genotype_params* p, int64_t n_sites, const int64_t* site_keys, const uint8_t* alt_sym,
                 uint32_t* dp, uint32_t* alt, int on_device) {
    if (c->n_contigs <= 0) { set_error("lsg_genotype_cells: no contigs set"); return -2; }
    if (c->n_cb <= 0) { set_error("lsg_genotype_cells: no barcodes set"); return -2; }
    if (n_sites < 0 || (n_sites > 0 && (!site_keys || !alt_sym || !dp || !alt))) { set_error("lsg_genotype_cells: bad arguments"); return -2; }
    if (n_sites == 0) return 0;
    if (!on_device)
        for (int64_t i = 1; i < n_sites; ++i)
            if (site_keys[i] <= site_keys[i - 1]) { set_error("lsg_genotype_cells: site keys must be strictly ascending"); return -2; }
    hipStream_t st = c->stream;
    const size_t cells = (size_t)n_sites * (size_t)c->n_cb;
    DevBuf d_keys, d_alt_sym, d_dp, d_alt, d_cbk;
    auto done = [&](int rc) { d_keys.release(); d_alt_sym.release(); d_dp.release(); d_alt.release(); d_cbk.release(); return rc; };
    GenoArgs a{};
    a.n_reads = c->rd.n_reads; a.n_segs = c->rd.n_segs;
    a.read_tid = c->rd.read_tid; a.read_flag = c->rd.read_flag; a.read_mapq = c->rd.read_mapq; a.read_cb = c->rd.read_cb;
    a.seg_read = c->rd.seg_read; a.seg_start = c->rd.seg_start; a.seg_len = c->rd.seg_len; a.seg_ev_off = c->rd.seg_ev_off;
    a.events = c->rd.events;
    a.celltype_of = c->d_celltype_of.as<uint8_t>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.p = *p; a.n_sites = n_sites;
    if (d_cbk.reserve((size_t)(a.n_reads + 1) * 4)) return done(-1);
    a.read_cbk = d_cbk.as<uint32_t>();
    if (on_device) { a.site_keys = site_keys; a.alt_sym = alt_sym; a.dp = dp; a.alt = alt; }
    else {
        if (d_keys.reserve((size_t)n_sites * 8) || d_alt_sym.reserve((size_t)n_sites) || d_dp.reserve(cells * 4) || d_alt.reserve(cells * 4)) return done(-1);
        if (hipMemcpyAsync(d_keys.p, site_keys, (size_t)n_sites * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_alt_sym.p, alt_sym, (size_t)n_sites, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("lsg_genotype_cells: upload failed"); return done(-1); }
        a.site_keys = d_keys.as<int64_t>(); a.alt_sym = d_alt_sym.as<uint8_t>(); a.dp = d_dp.as<uint32_t>(); a.alt = d_alt.as<uint32_t>();
    }
    if (hipMemsetAsync(a.dp, 0, cells * 4, st) != hipSuccess || hipMemsetAsync(a.alt, 0, cells * 4, st) != hipSuccess) { set_error("lsg_genotype_cells: memset failed"); return done(-1); }
    const unsigned cap = (unsigned)(c->n_cus * 16);
    if (a.n_reads > 0) {
        unsigned g = (unsigned)((a.n_reads + 255) / 256); if (g > cap) g = cap;
        hipLaunchKernelGGL(k_geno_read_key, dim3(g), dim3(256), 0, st, a);
    }
    if (a.n_segs > 0) {
        unsigned g = (unsigned)((a.n_segs + 255) / 256); if (g > cap) g = cap;
        hipLaunchKernelGGL(k_geno_segments, dim3(g), dim3(256), 0, st, a);
    }
    if (!on_device) {
        if (hipMemcpyAsync(dp, a.dp, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(alt, a.alt, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("lsg_genotype_cells: download failed"); return done(-1); }
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("lsg_genotype_cells: kernel failed"); return done(-1); }
    return done(0);
}

} // namespace lsg
