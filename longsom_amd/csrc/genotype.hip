// Per-cell genotyping at target sites: HCCVSingleCellGenotype.py:82-220 and its twin
// SNVCalling/SingleCellGenotype.py:84-228.  A target site is one position of one tile of the store: a workgroup per site walks the
// tile's blocks, a thread takes the 16-byte row of the site's position (eight entries' events there) and adds every entry that has a
// countable event to the site's per-barcode (Dp, Alt) pair.  Target sites are a few thousand; the pass reads the targets' tiles only.
#include "lsg_ctx.h"

namespace lsg {

struct GenoArgs {
    const uint4* store; const uint16_t* ext; const uint32_t* s0; const uint32_t* fm;
    const uint32_t* tile_base; const uint32_t* tile_off; const uint32_t* blk_off;
    const uint8_t* celltype_of; const int64_t* contig_len;
    int32_t n_contigs, n_cb;
    lsg_genotype_params p;
    int64_t n_sites; const int64_t* site_keys; const uint8_t* alt_sym;
    uint32_t* dp; uint32_t* alt;
};

// entry admission: pileup flag filter + ignore_orphans + min_mq (HCCVSingleCellGenotype.py:123), not secondary /
// duplicate / supplementary (:168), CB present and in barcodes.tsv (:160-164)
__global__ __launch_bounds__(256) void k_geno_sites(GenoArgs a) {
    const int64_t i = blockIdx.x;
    const int64_t key = a.site_keys[i];
    const int64_t tid = key >> 32, pos = key & 0xffffffffll;
    if (tid < 0 || tid >= a.n_contigs || pos >= a.contig_len[tid]) return;
    const uint32_t t = a.tile_base[tid] + (uint32_t)(pos >> 6), q = (uint32_t)pos & 63u;
    const uint32_t n = a.tile_off[t + 1] - a.tile_off[t], b0 = a.blk_off[t];
    const uint32_t alt_sym = a.alt_sym[i];
    for (uint32_t k = threadIdx.x; k < (n + 7u) / 8u; k += blockDim.x) {
        const uint32_t x = a.ext[b0 + k];                                          // rows outside the block's extent hold no event (and were never written)
        if (q < (x & 0xffu) || q >= (x >> 8)) continue;
        const uint4 row = a.store[(uint64_t)(b0 + k) * 64 + q];
        const uint32_t w[4] = {row.x, row.y, row.z, row.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t ev = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
            if (!(ev & LSG_EVENT_VALID) || (int)(ev & 0xffu) < a.p.min_bq) continue;
            const uint32_t sym = (ev >> 8) & 7u;
            if (sym > (uint32_t)LSG_SYM_N) continue;                               // 'O' is not in Bases (:148)
            const bool is_alt = sym == alt_sym;
            if (a.p.alt_only && !is_alt) continue;
            const uint64_t p = (uint64_t)(b0 + k) * 8 + u;
            const uint32_t cb = a.s0[p] & CB_MASK, f = a.fm[p], flag = f & 0xffffu;
            bool ok = cb < (uint32_t)a.n_cb && (flag & a.p.flag_exclude) == 0 && (int)(f >> 16) >= a.p.min_mq;
            if (ok && a.p.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) ok = false;
            if (ok && a.p.strict_cb && (flag & LSG_FLAG_CB_SUFFIX)) ok = false;
            if (!ok || a.celltype_of[cb] == 255) continue;
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
    DevBuf d_keys, d_alt_sym, d_dp, d_alt;
    auto done = [&](int rc) { d_keys.release(); d_alt_sym.release(); d_dp.release(); d_alt.release(); return rc; };
    if (!c->tm_valid) { set_error("lsg_genotype_cells: no reads loaded"); return -2; }
    GenoArgs a{};
    a.store = c->tm[TM_STORE].as<uint4>(); a.ext = c->tm[TM_EXT].as<uint16_t>(); a.s0 = c->tm[TM_S0].as<uint32_t>(); a.fm = c->tm[TM_FM].as<uint32_t>();
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.tile_off = c->d_tile_off.as<uint32_t>(); a.blk_off = c->tm[TM_BLK_OFF].as<uint32_t>();
    a.celltype_of = c->d_celltype_of.as<uint8_t>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.p = *p; a.n_sites = n_sites;
    if (on_device) { a.site_keys = site_keys; a.alt_sym = alt_sym; a.dp = dp; a.alt = alt; }
    else {
        if (d_keys.reserve((size_t)n_sites * 8) || d_alt_sym.reserve((size_t)n_sites) || d_dp.reserve(cells * 4) || d_alt.reserve(cells * 4)) return done(-1);
        if (hipMemcpyAsync(d_keys.p, site_keys, (size_t)n_sites * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_alt_sym.p, alt_sym, (size_t)n_sites, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("lsg_genotype_cells: upload failed"); return done(-1); }
        a.site_keys = d_keys.as<int64_t>(); a.alt_sym = d_alt_sym.as<uint8_t>(); a.dp = d_dp.as<uint32_t>(); a.alt = d_alt.as<uint32_t>();
    }
    if (hipMemsetAsync(a.dp, 0, cells * 4, st) != hipSuccess || hipMemsetAsync(a.alt, 0, cells * 4, st) != hipSuccess) { set_error("lsg_genotype_cells: memset failed"); return done(-1); }
    if (c->tm_nblk) hipLaunchKernelGGL(k_geno_sites, dim3((unsigned)n_sites), dim3(256), 0, st, a);
    if (!on_device) {
        if (hipMemcpyAsync(dp, a.dp, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(alt, a.alt, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("lsg_genotype_cells: download failed"); return done(-1); }
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("lsg_genotype_cells: kernel failed"); return done(-1); }
    return done(0);
}

} // namespace lsg
