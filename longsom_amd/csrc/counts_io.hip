// lsg_load_counts: install per-cell-type count rows (e.g. parsed from BaseCellCounter TSVs) as if
// lsg_pileup_count had produced them, so that lsg_call_step1 can serve the MergeCounts /
// BaseCellCalling_step1 rules on their own file inputs (R:SNVCalling.smk:62-156).
#include "lsg_ctx.h"
#include <algorithm>

namespace lsg {
int install_counts(lsg_ctx* c, int32_t n_ct, const int64_t* const* keys, const uint32_t* const* counts, const int64_t* n_rows) {
    if (c->n_contigs <= 0) { set_error("lsg_load_counts: set contigs first"); return -2; }
    if (n_ct <= 0 || n_ct > LSG_MAX_CELLTYPES) { set_error("lsg_load_counts: n_celltypes %d not in [1,%d]", n_ct, LSG_MAX_CELLTYPES); return -2; }
    struct Unit { uint32_t u; uint64_t mask; uint32_t rowbase; int32_t tstart; int32_t tid; int32_t ct; };
    std::vector<Unit> units;
    uint64_t max_rows = 1;
    for (int ct = 0; ct < n_ct; ++ct) {
        const int64_t n = n_rows[ct];
        if ((uint64_t)n > max_rows) max_rows = (uint64_t)n;
        int64_t prev = -1;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t k = keys[ct][i];
            if (k <= prev) { set_error("lsg_load_counts: rows of cell type %d are not strictly sorted at row %lld", ct, (long long)i); return -2; }
            prev = k;
            const int32_t tid = (int32_t)(k >> 32);
            const int64_t pos = k & 0xffffffffll;
            if (tid < 0 || tid >= c->n_contigs || pos >= c->contig_len[tid]) { set_error("lsg_load_counts: site outside the contig table (tid %d pos %lld)", tid, (long long)pos); return -2; }
            const uint32_t tile = c->tile_base[tid] + (uint32_t)(pos >> 6);
            const uint32_t u = tile * (uint32_t)n_ct + (uint32_t)ct;
            if (units.empty() || units.back().u != u || units.back().ct != ct)
                units.push_back(Unit{u, 0, (uint32_t)i, (int32_t)((pos >> 6) << 6), tid, ct});
            units.back().mask |= 1ull << (pos & 63);
        }
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.u < b.u; });
    const uint32_t n_ne = (uint32_t)units.size();
    std::vector<uint32_t> ne_units(n_ne + 1), ne_rowbase(n_ne + 1);
    std::vector<uint64_t> ne_mask(n_ne + 1);
    std::vector<int2> ne_geom(n_ne + 1);
    for (uint32_t w = 0; w < n_ne; ++w) {
        ne_units[w] = units[w].u; ne_rowbase[w] = units[w].rowbase; ne_mask[w] = units[w].mask;
        ne_geom[w] = make_int2(units[w].tstart, units[w].tid | (units[w].ct << 24));
    }
    hipStream_t st = c->stream;
    if (c->d_ne_units.reserve((size_t)(n_ne + 2) * 4) || c->d_ne_mask.reserve((size_t)(n_ne + 2) * 8) ||
        c->d_ne_rowbase.reserve((size_t)(n_ne + 2) * 4) || c->ws[WS_NE_GEOM].reserve((size_t)(n_ne + 2) * 8)) return -1;
    if (n_ne) {
        LSG_HIP(hipMemcpyAsync(c->d_ne_units.p, ne_units.data(), (size_t)n_ne * 4, hipMemcpyHostToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->d_ne_mask.p, ne_mask.data(), (size_t)n_ne * 8, hipMemcpyHostToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->d_ne_rowbase.p, ne_rowbase.data(), (size_t)n_ne * 4, hipMemcpyHostToDevice, st));
        LSG_HIP(hipMemcpyAsync(c->ws[WS_NE_GEOM].p, ne_geom.data(), (size_t)n_ne * 8, hipMemcpyHostToDevice, st));
    }
    // blocked planes (lsg::row_word; the reverse-strand plane is derived: BCr = BC - BCf)
    max_rows = (max_rows + 63) / 64 * 64;
    c->row_cap = max_rows;
    std::vector<uint32_t> planes((size_t)max_rows * lsg::ROW_STORED_WORDS, 0u);
    for (int ct = 0; ct < n_ct; ++ct) {
        const int64_t n = n_rows[ct];
        if (c->d_rows[ct].reserve((size_t)max_rows * lsg::ROW_STORED_WORDS * 4)) return -1;
        for (int64_t i = 0; i < n; ++i)
            for (int k = 0; k < lsg::ROW_PLANES; ++k) planes[(size_t)lsg::row_word((uint64_t)i, k)] = counts[ct][i * LSG_ROW_WORDS + k];
        if (n) LSG_HIP(hipMemcpyAsync(c->d_rows[ct].p, planes.data(), (size_t)max_rows * lsg::ROW_STORED_WORDS * 4, hipMemcpyHostToDevice, st));
        LSG_HIP(hipStreamSynchronize(st));
        c->n_rows[ct] = n;
    }
    LSG_HIP(hipStreamSynchronize(st));
    c->n_ct = n_ct;
    c->n_ne = n_ne;
    c->n_columns = 0;
    c->last_params = lsg_count_params{};
    c->counted = true;
    ++c->count_serial;
    c->called = false;
    return 0;
}
} // namespace lsg

extern "C" int lsg_load_counts(lsg_ctx* c, int32_t n_celltypes, const int64_t* const* keys, const uint32_t* const* counts, const int64_t* n_rows) {
    if (!c || !keys || !counts || !n_rows) { lsg::set_error("lsg_load_counts: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return lsg::install_counts(c, n_celltypes, keys, counts, n_rows);
}
