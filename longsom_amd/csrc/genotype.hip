// Per-cell genotyping at target sites: HCCVSingleCellGenotype.py:82-220 and its twin
// SNVCalling/SingleCellGenotype.py:84-228.  A target site is one position of one tile of the store: a workgroup per site walks the
// tile's blocks, a thread takes the 16-byte row of the site's position (eight entries' events there) and adds every entry that has a
// countable event to the site's per-barcode (Dp, Alt) pair.  Target sites are a few thousand; the pass reads the targets' tiles only.
#include "lsg_ctx.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace lsg {

struct GenoArgs {
    const uint4* store; const uint16_t* ext; const uint32_t* s0; const uint16_t* read_flag; const uint8_t* read_mapq;
    const uint32_t* tile_base; const uint32_t* tile_off; const uint32_t* blk_off;
    const uint8_t* celltype_of; const int64_t* contig_len;
    int32_t n_contigs, n_cb;
    lsg_genotype_params p;
    int64_t n_sites; const int64_t* site_keys; const uint8_t* alt_sym;
    uint32_t* dp; uint32_t* alt;
    int64_t site0;                        // the launch covers the sites [site0, site0 + gridDim.x)
    const uint32_t* rd; const uint8_t* read_drop;      // the entries' reads; reads the pileup's max_depth rule dropped for this launch's sites (or null)
};

// entry admission: pileup flag filter + ignore_orphans + min_mq (HCCVSingleCellGenotype.py:123), not secondary /
// duplicate / supplementary (:168), CB present and in barcodes.tsv (:160-164)
__global__ __launch_bounds__(256) void k_geno_sites(GenoArgs a) {
    const int64_t i = a.site0 + blockIdx.x;
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
            const uint32_t cb = a.s0[p] & CB_MASK;
            if (cb >= (uint32_t)a.n_cb || a.celltype_of[cb] == 255) continue;       // (pad entries carry CB_MASK)
            const uint32_t r = a.rd[p], flag = a.read_flag[r];                       // the entry's read: its SAM flag (LSG_FLAG_CB_SUFFIX included) and MAPQ
            bool ok = (flag & a.p.flag_exclude) == 0 && (int)a.read_mapq[r] >= a.p.min_mq;
            if (ok && a.p.ignore_orphans && (flag & 0x1) && !(flag & 0x2)) ok = false;
            if (ok && a.p.strict_cb && (flag & LSG_FLAG_CB_SUFFIX)) ok = false;
            if (!ok) continue;
            if (a.read_drop && a.read_drop[r]) continue;
            const uint64_t cell = (uint64_t)i * (uint64_t)a.n_cb + cb;
            atomicAdd(&a.dp[cell], 1u);
            if (is_alt) atomicAdd(&a.alt[cell], 1u);
        }
    }
}

// The reads of the resident load on the host, in coordinate order, for the replay of the pileup's max_depth rule over a REGION
// (HCCVSingleCellGenotype.py:122: bam.pileup(CHROM, START, END, ..., max_depth = 200000) per window of target sites): the reads that
// overlap [start, end) pass through htslib's buffer in file order; a read that is not the first of its start position is dropped while
// the buffer (reads that entered and end at or after that position, + 1) exceeds max_depth (layout.hip depth_cap_drops states the rule).
struct GenoHostReads {
    std::vector<int32_t> tid, pos, end; std::vector<uint16_t> flag; std::vector<uint8_t> mapq;
    std::vector<uint32_t> order;          // coordinate order (stable)
    std::vector<int32_t> pmax;            // running maximum of `end` along `order`, restarted at every contig
    int fetch(lsg_ctx* c) {
        hipStream_t st = c->stream;
        const int64_t R = c->rd.n_reads, S = c->rd.n_segs;
        DevBuf d_end;
        if (d_end.reserve((size_t)(R > 0 ? R : 1) * 4)) return -1;
        hipLaunchKernelGGL(k_read_end_init, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, c->rd.read_pos, R, d_end.as<int32_t>());
        if (S > 0) hipLaunchKernelGGL(k_read_end, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, c->rd.seg_read, c->rd.seg_start, c->rd.seg_len, S, d_end.as<int32_t>());
        tid.resize((size_t)R); pos.resize((size_t)R); end.resize((size_t)R); flag.resize((size_t)R); mapq.resize((size_t)R);
        auto cp = [&](void* dst, const void* src, size_t n) { return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st) != hipSuccess; };
        const bool bad = cp(tid.data(), c->rd.read_tid, (size_t)R * 4) || cp(pos.data(), c->rd.read_pos, (size_t)R * 4) || cp(end.data(), d_end.p, (size_t)R * 4) ||
                         cp(flag.data(), c->rd.read_flag, (size_t)R * 2) || cp(mapq.data(), c->rd.read_mapq, (size_t)R) || hipStreamSynchronize(st) != hipSuccess;
        d_end.release();
        if (bad) { set_error("lsg_genotype_cells: copying the reads for the depth cap failed"); return -1; }
        order.resize((size_t)R);
        for (int64_t i = 0; i < R; ++i) order[(size_t)i] = (uint32_t)i;
        auto key = [&](uint32_t i) { return ((uint64_t)(uint32_t)tid[i] << 32) | (uint32_t)pos[i]; };
        bool sorted = true;
        for (int64_t i = 1; i < R && sorted; ++i) sorted = key((uint32_t)(i - 1)) <= key((uint32_t)i);
        if (!sorted) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        pmax.resize((size_t)R);
        for (int64_t k = 0; k < R; ++k) {
            const uint32_t i = order[(size_t)k];
            const bool fresh = k == 0 || tid[order[(size_t)k - 1]] != tid[i];
            pmax[(size_t)k] = fresh ? end[i] : std::max(pmax[(size_t)k - 1], end[i]);
        }
        return 0;
    }
    // marks drop[i] = 1 for the reads the rule drops in the pileup of (t, [start, end_)); `touched` lists them (the caller clears them again)
    void replay(int32_t t, int64_t start, int64_t end_, const lsg_genotype_params& p, int max_depth, std::vector<uint8_t>& drop, std::vector<uint32_t>& touched) const {
        const size_t R = order.size();
        auto tkey = [&](size_t k) { return tid[order[k]]; };
        size_t lo = 0, hi = R;                                   // first k of contig t
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (tkey(mid) < t) lo = mid + 1; else hi = mid; }
        const size_t k0 = lo;
        hi = R;                                                  // first k of contig t whose read starts at or after end_ (or of a later contig)
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (tkey(mid) < t || (tkey(mid) == t && (int64_t)pos[order[mid]] < end_)) lo = mid + 1; else hi = mid; }
        const size_t k1 = lo;
        lo = k0; hi = k1;                                        // first k whose running maximum of the ends passes `start`: nothing before it overlaps the region
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((int64_t)pmax[mid] <= start) lo = mid + 1; else hi = mid; }
        const uint32_t pool_flags = p.flag_exclude & ~0x800u;     // (supplementary reads pass the pileup's filter: only :168 drops them later)
        std::vector<int32_t> heap;
        auto cmp = [](int32_t a, int32_t b) { return a > b; };
        int32_t cur_pos = -1; bool first_here = true;
        for (size_t k = lo; k < k1; ++k) {
            const uint32_t i = order[k];
            if ((int64_t)end[i] <= start) continue;               // not in the region: never fetched
            if ((int)mapq[i] < p.min_mq || (flag[i] & pool_flags)) continue;
            if (p.ignore_orphans && (flag[i] & 1u) && !(flag[i] & 2u)) continue;
            if (pos[i] != cur_pos) {
                cur_pos = pos[i]; first_here = true;
                while (!heap.empty() && heap.front() < cur_pos) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.pop_back(); }
            }
            if (!first_here && (int64_t)heap.size() + 1 > (int64_t)max_depth) { drop[i] = 1; touched.push_back(i); continue; }
            first_here = false;
            heap.push_back(end[i]); std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
};

// n_groups > 0: the sites [group_off[g], group_off[g + 1]) are one pileup call of the reference (a window of its target sites, region
// [first site - 1, last site + 1)): with max_depth > 0 its depth cap is replayed per group.  n_groups == 0: no cap.
int run_genotype(lsg_ctx* c, const lsg_genotype_params* p, int64_t n_sites, const int64_t* site_keys, const uint8_t* alt_sym,
                 uint32_t* dp, uint32_t* alt, int on_device, int32_t max_depth, int64_t n_groups, const int64_t* group_off) {
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
    if (c->store_skipped) { set_error("lsg_genotype_cells: the load kept no store (lsg_set_store_policy): load the reads again with LSG_STORE_KEEP"); return -2; }
    if (!c->tm_valid) { set_error("lsg_genotype_cells: no reads loaded"); return -2; }
    GenoArgs a{};
    a.store = c->tm[TM_STORE].as<uint4>(); a.ext = c->tm[TM_EXT].as<uint16_t>(); a.s0 = c->tm[TM_S0].as<uint32_t>(); a.read_flag = c->rd.read_flag; a.read_mapq = c->rd.read_mapq;
    a.tile_base = c->d_tile_base.as<uint32_t>(); a.tile_off = c->d_tile_off.as<uint32_t>(); a.blk_off = c->tm[TM_BLK_OFF].as<uint32_t>();
    a.celltype_of = c->d_celltype_of.as<uint8_t>(); a.contig_len = c->d_contig_len.as<int64_t>();
    a.n_contigs = c->n_contigs; a.n_cb = c->n_cb; a.p = *p; a.n_sites = n_sites;
    a.site0 = 0; a.rd = c->tm[TM_RD].as<uint32_t>(); a.read_drop = nullptr;
    if (on_device) { a.site_keys = site_keys; a.alt_sym = alt_sym; a.dp = dp; a.alt = alt; }
    else {
        if (d_keys.reserve((size_t)n_sites * 8) || d_alt_sym.reserve((size_t)n_sites) || d_dp.reserve(cells * 4) || d_alt.reserve(cells * 4)) return done(-1);
        if (hipMemcpyAsync(d_keys.p, site_keys, (size_t)n_sites * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_alt_sym.p, alt_sym, (size_t)n_sites, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("lsg_genotype_cells: upload failed"); return done(-1); }
        a.site_keys = d_keys.as<int64_t>(); a.alt_sym = d_alt_sym.as<uint8_t>(); a.dp = d_dp.as<uint32_t>(); a.alt = d_alt.as<uint32_t>();
    }
    if (hipMemsetAsync(a.dp, 0, cells * 4, st) != hipSuccess || hipMemsetAsync(a.alt, 0, cells * 4, st) != hipSuccess) { set_error("lsg_genotype_cells: memset failed"); return done(-1); }
    bool capped = false;
    if (c->tm_nblk && max_depth > 0 && n_groups > 0) {
        if (live_read_bound_all(c)) return done(-1);
        capped = c->max_live_all + 1 > (int64_t)max_depth;       // (otherwise not even all resident reads together fill a buffer)
    }
    if (c->tm_nblk && !capped) hipLaunchKernelGGL(k_geno_sites, dim3((unsigned)n_sites), dim3(256), 0, st, a);
    if (c->tm_nblk && capped) {
        std::vector<int64_t> h_keys((size_t)n_sites);
        if (on_device) { if (hipMemcpyAsync(h_keys.data(), site_keys, (size_t)n_sites * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { set_error("lsg_genotype_cells: download failed"); return done(-1); } }
        else memcpy(h_keys.data(), site_keys, (size_t)n_sites * 8);
        GenoHostReads hr;
        if (hr.fetch(c)) return done(-1);
        const int64_t R = c->rd.n_reads;
        std::vector<uint8_t> drop((size_t)(R > 0 ? R : 1), 0);
        std::vector<uint32_t> touched;
        DevBuf d_drop;
        if (d_drop.reserve((size_t)(R > 0 ? R : 1))) return done(-1);
        auto done2 = [&](int rc) { d_drop.release(); return done(rc); };
        for (int64_t g = 0; g < n_groups; ++g) {
            const int64_t s0 = group_off[g], s1 = group_off[g + 1];
            if (s0 < 0 || s1 < s0 || s1 > n_sites) { set_error("lsg_genotype_cells: group offsets must be ascending and end at the number of sites"); return done2(-2); }
            if (s1 == s0) continue;
            const int64_t t = h_keys[(size_t)s0] >> 32;
            if ((h_keys[(size_t)s1 - 1] >> 32) != t) { set_error("lsg_genotype_cells: a group of sites spans two contigs"); return done2(-2); }
            touched.clear();
            // the region of the group's pileup (HCCVSingleCellGenotype.py:109-110,122): [first site - 1, last site + 1), 0-based
            hr.replay((int32_t)t, (h_keys[(size_t)s0] & 0xffffffffll) - 1, (h_keys[(size_t)s1 - 1] & 0xffffffffll) + 1, *p, max_depth, drop, touched);
            a.site0 = s0; a.read_drop = nullptr;
            if (!touched.empty()) {
                if (hipMemcpyAsync(d_drop.p, drop.data(), (size_t)R, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("lsg_genotype_cells: upload failed"); return done2(-1); }
                a.read_drop = d_drop.as<uint8_t>();
            }
            hipLaunchKernelGGL(k_geno_sites, dim3((unsigned)(s1 - s0)), dim3(256), 0, st, a);
            if (!touched.empty()) {
                if (hipStreamSynchronize(st) != hipSuccess) { set_error("lsg_genotype_cells: kernel failed"); return done2(-1); }      // (the mask is reused by the next group)
                for (uint32_t i : touched) drop[i] = 0;
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("lsg_genotype_cells: kernel failed"); return done2(-1); }
        d_drop.release();
    }
    if (!on_device) {
        if (hipMemcpyAsync(dp, a.dp, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(alt, a.alt, cells * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("lsg_genotype_cells: download failed"); return done(-1); }
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_error("lsg_genotype_cells: kernel failed"); return done(-1); }
    return done(0);
}

} // namespace lsg
