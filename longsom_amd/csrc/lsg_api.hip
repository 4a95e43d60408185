// C-ABI entry points of liblongsom_hip.so (see include/longsom_hip.h).
#include "lsg_ctx.h"
#include <cstdarg>
#include <cstring>

namespace lsg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* get_error() { return g_err; }

int run_count(lsg_ctx* c, const lsg_count_params* p);
int run_fetch_counts(lsg_ctx* c, int ct, int64_t* keys, uint8_t* ref, uint32_t* counts, int64_t capacity);
int run_call(lsg_ctx* c, const lsg_call_params* p);
int run_fetch_calls(lsg_ctx* c, lsg_call* out, int64_t capacity, int candidates_only, int64_t* n_out);
int run_select_calls(lsg_ctx* c, int kind, lsg_call* dst_device, int64_t capacity, int64_t* n_out);
int run_probe(lsg_ctx* c, int kind, const int64_t* keys, int64_t n, uint8_t* hits, int on_device);
int run_set_table_names(lsg_ctx* c, int32_t n_contigs, const char* contig_names, int32_t n_ct, const char* ct_names);
int run_format_table(lsg_ctx* c, int32_t table, int64_t* n_bytes);
int run_copy_table(lsg_ctx* c, int32_t table, char* dst, int64_t capacity);
int run_append_table(lsg_ctx* c, int32_t table, const char* path);
int run_free_table(lsg_ctx* c, int32_t table);
int run_step2_summary(lsg_ctx* c, int32_t n_cols, uint8_t* kinds, int64_t* n_survivor_bytes);
int run_genotype(lsg_ctx* c, const lsg_genotype_params* p, int64_t n_sites, const int64_t* site_keys, const uint8_t* alt_sym,
                 uint32_t* dp, uint32_t* alt, int on_device, int32_t max_depth, int64_t n_groups, const int64_t* group_off);
int run_sf4(lsg_ctx* c, int64_t items, const uint32_t* k, const uint32_t* n, double al, double be, int32_t* out, double* raw);

// copy a host or device array into a grow-only device buffer of the handle (the caller's array is free again when the call returns)
template <class T>
static int put(lsg_ctx* c, DevBuf& buf, const T*& dst, const T* src, int64_t n, int on_device, hipStream_t st = nullptr) {
    if (n <= 0 || !src) { dst = nullptr; return 0; }
    if (src == buf.as<T>()) { dst = src; return 0; }          // (lsg_synth_reads generates into the handle's own buffers)
    if (buf.reserve((size_t)n * sizeof(T))) return -1;
    LSG_HIP(hipMemcpyAsync(buf.p, src, (size_t)n * sizeof(T), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st ? st : c->stream));
    dst = buf.as<T>();
    return 0;
}
// a reference's bases: host arrays are copied, device arrays adopted
static int put_ref(lsg_ctx* c, DevBuf& buf, const uint8_t*& dst, const uint8_t* src, int64_t n, int on_device) {
    if (on_device) { dst = src; return 0; }
    if (buf.reserve((size_t)n)) return -1;
    LSG_HIP(hipMemcpyAsync(buf.p, src, (size_t)n, hipMemcpyHostToDevice, c->stream));
    dst = buf.as<uint8_t>();
    return 0;
}
} // namespace lsg

using namespace lsg;

extern "C" {

const char* lsg_last_error(void) { return get_error(); }
const char* lsg_version(void) { return "longsom_hip 0.1 (gfx950)"; }

int lsg_create(int device_id, lsg_ctx** out) {
    if (!out) { set_error("lsg_create: out is NULL"); return -2; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("lsg_create: no HIP device visible (this library has no CPU fallback)");
        return -1;
    }
    if (device_id < 0 || device_id >= n) { set_error("lsg_create: device %d out of range [0,%d)", device_id, n); return -2; }
    LSG_HIP(hipSetDevice(device_id));
    lsg_ctx* c = new lsg_ctx();
    c->device = device_id;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_blk, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_lpt, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("lsg_create: hipStreamCreate failed"); delete c; return -1;
    }
    c->stream = c->own_stream;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), 4096, hipHostMallocDefault) != hipSuccess) { set_error("lsg_create: hipHostMalloc failed"); delete c; return -1; }
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) { set_error("lsg_create: hipEventCreate failed"); delete c; return -1; }
    for (auto& e : c->evb)
        if (hipEventCreate(&e) != hipSuccess) { set_error("lsg_create: hipEventCreate failed"); delete c; return -1; }
    *out = c;
    return 0;
}

void lsg_destroy(lsg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_tile_base, &c->d_contig_len, &c->d_ref_ptrs, &c->d_celltype_of, &c->d_ct_rank, &c->b_read_tid, &c->b_read_pos,
                      &c->b_read_flag, &c->b_read_mapq, &c->b_read_cb, &c->b_seg_read, &c->b_seg_start, &c->b_seg_len,
                      &c->b_seg_ev_off, &c->b_events, &c->d_read_key, &c->d_read_drop, &c->d_drop_pairs, &c->d_read_adm,
                      &c->d_ne_units, &c->d_ne_mask, &c->d_ne_rowbase, &c->d_ne_rowoff,
                      &c->d_scalars, &c->d_cub_tmp, &c->d_ix_stat, &c->d_tile_cap, &c->d_tile_off, &c->d_calls, &c->d_site_off, &c->d_tail_table, &c->d_pass_list, &c->d_defer_list, &c->d_xcd_queues};
    for (auto* b : bufs) b->release();
    for (auto& b : c->d_rows) b.release();
    for (auto& b : c->ref) b.release();
    for (auto& s : c->posset) s.keys.release();
    for (auto& b : c->syn) b.release();
    for (auto& b : c->gen) b.release();
    (void)run_free_table(c, -1);
    c->tab_names.release();
    for (auto& b : c->ws) b.release();
    for (auto& b : c->tm) b.release();
    for (auto& b : c->bt) b.release();
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->evb) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->ev_blk) (void)hipEventDestroy(c->ev_blk);
    if (c->ev_lpt) (void)hipEventDestroy(c->ev_lpt);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    delete c;
}

int lsg_unload_reads(lsg_ctx* c) {
    if (!c) { set_error("lsg_unload_reads: NULL handle"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    LSG_HIP(hipStreamSynchronize(c->stream));
    DevBuf* bufs[] = {&c->b_read_tid, &c->b_read_pos, &c->b_read_flag, &c->b_read_mapq, &c->b_read_cb, &c->b_seg_read, &c->b_seg_start, &c->b_seg_len,
                      &c->b_seg_ev_off, &c->b_events, &c->d_read_key, &c->d_read_drop, &c->d_read_adm, &c->d_ne_units, &c->d_ne_mask, &c->d_ne_rowbase, &c->d_ne_rowoff,
                      &c->d_cub_tmp, &c->d_ix_stat, &c->d_tile_cap, &c->d_tile_off, &c->d_calls, &c->d_site_off, &c->d_pass_list, &c->d_defer_list};
    for (auto* b : bufs) b->release();
    for (auto& b : c->d_rows) b.release();
    for (auto& b : c->gen) b.release();
    for (auto& b : c->ws) b.release();
    for (auto& b : c->tm) b.release();
    for (auto& b : c->bt) b.release();
    c->rd = lsg_reads{};
    lsg::drop_store(c);
    c->row_cap = 0; c->n_ne = 0; c->n_columns = 0; c->n_sites = 0; c->n_cand = 0; c->n_pass = -1;
    for (auto& n : c->n_rows) n = 0;
    return 0;
}

int lsg_set_stream(lsg_ctx* c, void* hip_stream) {
    if (!c) { set_error("lsg_set_stream: NULL handle"); return -2; }
    (void)hipStreamSynchronize(c->stream);
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return 0;
}

int lsg_synchronize(lsg_ctx* c) {
    if (!c) { set_error("lsg_synchronize: NULL handle"); return -2; }
    LSG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int lsg_set_contigs(lsg_ctx* c, int32_t n_contigs, const int64_t* lengths) {
    if (!c || n_contigs <= 0 || !lengths) { set_error("lsg_set_contigs: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    c->n_contigs = n_contigs;
    c->tab_n_contigs = 0;                 // (the names the tables print belong to the contig table they were set for)
    c->contig_len.assign(lengths, lengths + n_contigs);
    c->tile_base.assign(n_contigs + 1, 0);
    uint64_t t = 0;
    for (int i = 0; i < n_contigs; ++i) {
        if (lengths[i] < 0 || lengths[i] > 0x7fffffffll) { set_error("lsg_set_contigs: contig %d length %lld unsupported", i, (long long)lengths[i]); return -2; }
        c->tile_base[i] = (uint32_t)t;
        // (an EVEN number of tiles per contig: a 128-position window - store.hip's bins of a load that keeps no store - is then tiles
        // (2 w, 2 w + 1) of the same contig everywhere; a contig's odd last tile is followed by one that holds no position)
        t += (uint64_t)(((lengths[i] + TILE_W - 1) / TILE_W + 1) & ~1ll);
    }
    if (t * LSG_MAX_CELLTYPES >= 0x7fffffffull) { set_error("lsg_set_contigs: genome too large (%llu tiles)", (unsigned long long)t); return -2; }
    c->tile_base[n_contigs] = (uint32_t)t;
    c->n_tiles = (uint32_t)t;
    c->tile_lo = 0; c->tile_hi = (uint32_t)t;
    lsg::drop_store(c);                        // the store's tiles are the contigs' tiles: reads are loaded after the contigs
    c->rd = lsg_reads{};
    for (auto& b : c->ref) b.release();
    c->ref.assign(n_contigs, DevBuf());
    c->ref_ptr.assign(n_contigs, nullptr);
    if (c->d_tile_base.reserve((size_t)(n_contigs + 1) * 4) || c->d_contig_len.reserve((size_t)n_contigs * 8) ||
        c->d_ref_ptrs.reserve((size_t)n_contigs * sizeof(void*))) return -1;
    LSG_HIP(hipMemcpyAsync(c->d_tile_base.p, c->tile_base.data(), (size_t)(n_contigs + 1) * 4, hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipMemcpyAsync(c->d_contig_len.p, c->contig_len.data(), (size_t)n_contigs * 8, hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipMemsetAsync(c->d_ref_ptrs.p, 0, (size_t)n_contigs * sizeof(void*), c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    c->counted = c->called = false;
    return 0;
}

int lsg_load_reference(lsg_ctx* c, int32_t tid, const uint8_t* bases, int64_t len, int32_t on_device) {
    if (!c || !bases) { set_error("lsg_load_reference: bad arguments"); return -2; }
    if (tid < 0 || tid >= c->n_contigs) { set_error("lsg_load_reference: tid %d out of range", tid); return -2; }
    if (len != c->contig_len[tid]) { set_error("lsg_load_reference: contig %d has length %lld, got %lld bases", tid, (long long)c->contig_len[tid], (long long)len); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    const uint8_t* p = nullptr;
    if (put_ref(c, c->ref[tid], p, bases, len > 0 ? len : 1, on_device)) return -1;
    c->ref_ptr[tid] = p;
    LSG_HIP(hipMemcpyAsync(c->d_ref_ptrs.as<const uint8_t*>() + tid, &c->ref_ptr[tid], sizeof(void*), hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    c->counted = c->called = false;
    return 0;
}

int lsg_set_barcodes(lsg_ctx* c, const uint8_t* celltype_of, int32_t n_cb, int32_t n_celltypes) {
    if (!c || !celltype_of || n_cb <= 0) { set_error("lsg_set_barcodes: bad arguments"); return -2; }
    if (n_celltypes <= 0 || n_celltypes > LSG_MAX_CELLTYPES) { set_error("lsg_set_barcodes: n_celltypes %d not in [1,%d]", n_celltypes, LSG_MAX_CELLTYPES); return -2; }
    if (n_cb > 0x00FFFFFF) { set_error("lsg_set_barcodes: more than 2^24-1 barcodes"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    // the table that is already set stays as it is (setting one drops the resident count: the one a load made under lsg_set_count_at_load
    // would be made again) - decided HERE, against what the context holds (lsg_load_counts changes n_ct behind any caller's cache)
    if (c->n_cb == n_cb && c->n_ct == n_celltypes && c->h_celltype_of.size() == (size_t)n_cb && memcmp(c->h_celltype_of.data(), celltype_of, (size_t)n_cb) == 0) return 0;
    c->h_celltype_of.assign(celltype_of, celltype_of + n_cb);
    if (c->d_celltype_of.reserve((size_t)n_cb)) return -1;
    LSG_HIP(hipMemcpyAsync(c->d_celltype_of.p, celltype_of, (size_t)n_cb, hipMemcpyHostToDevice, c->stream));
    {
        std::vector<uint32_t> rank((size_t)n_cb, 0u);
        for (auto& v : c->ct_size) v = 0;
        for (int32_t i = 0; i < n_cb; ++i)
            if (celltype_of[i] < n_celltypes) rank[i] = c->ct_size[celltype_of[i]]++;
        if (c->d_ct_rank.reserve((size_t)n_cb * 4)) return -1;
        LSG_HIP(hipMemcpyAsync(c->d_ct_rank.p, rank.data(), (size_t)n_cb * 4, hipMemcpyHostToDevice, c->stream));
        LSG_HIP(hipStreamSynchronize(c->stream));
    }
    LSG_HIP(hipStreamSynchronize(c->stream));
    c->n_cb = n_cb; c->n_ct = n_celltypes;
    c->max_live_reads = -1; c->max_live_exact = -1;
    c->counted = c->called = false;
    return 0;
}

int lsg_set_keep_reads(lsg_ctx* c, int32_t keep) {
    if (!c) { set_error("lsg_set_keep_reads: NULL handle"); return -2; }
    c->keep_reads = keep != 0;
    return 0;
}

int lsg_set_events_layout(lsg_ctx* c, int32_t layout) {
    if (!c || (layout != LSG_LAYOUT_COMPACT && layout != LSG_LAYOUT_PHASED)) { set_error("lsg_set_events_layout: bad arguments"); return -2; }
    c->events_layout = layout;
    return 0;
}

int lsg_set_pileup_window(lsg_ctx* c, int32_t window) {
    if (!c || window < 64) { set_error("lsg_set_pileup_window: the window must be at least 64 positions"); return -2; }
    c->plp_window = window;
    return 0;
}

int lsg_set_count_at_load(lsg_ctx* c, const lsg_count_params* params) {
    if (!c) { set_error("lsg_set_count_at_load: NULL handle"); return -2; }
    c->cal_enabled = params != nullptr;
    if (params) c->cal_params = *params;
    return 0;
}

int lsg_set_keep_unlisted(lsg_ctx* c, int32_t on) {
    if (!c) { set_error("lsg_set_keep_unlisted: NULL handle"); return -2; }
    c->keep_unlisted = on != 0;
    return 0;
}

int lsg_set_store_policy(lsg_ctx* c, int32_t policy) {
    if (!c) { set_error("lsg_set_store_policy: NULL handle"); return -2; }
    if (policy != LSG_STORE_KEEP && policy != LSG_STORE_SKIP_WHEN_COUNTED) { set_error("lsg_set_store_policy: unknown policy %d", (int)policy); return -2; }
    c->store_policy = policy;
    return 0;
}

int lsg_set_load_filter(lsg_ctx* c, int32_t min_mq, uint32_t flag_exclude, int32_t ignore_orphans) {
    if (!c) { set_error("lsg_set_load_filter: NULL handle"); return -2; }
    c->lf_min_mq = min_mq; c->lf_flag_exclude = flag_exclude; c->lf_ignore_orphans = ignore_orphans ? 1 : 0;
    return 0;
}

// the resident store holds only reads that passed the load filter: a count (or a genotyping pass) that would admit more is refused
static int check_load_filter(lsg_ctx* c, const char* who, int32_t min_mq, uint32_t flag_exclude, int32_t ignore_orphans) {
    if (min_mq < c->st_min_mq || (c->st_flag_exclude & ~flag_exclude) != 0 || (c->st_ignore_orphans && !ignore_orphans)) {
        set_error("%s: these read filters (min_mq %d, flag_exclude 0x%x, ignore_orphans %d) admit reads the load filter dropped (min_mq %d, flag_exclude 0x%x, ignore_orphans %d): "
                  "load the reads again under a filter no stricter than the counts' (lsg_set_load_filter)", who, min_mq, flag_exclude, ignore_orphans,
                  c->st_min_mq, c->st_flag_exclude, c->st_ignore_orphans);
        return -2;
    }
    return 0;
}

int lsg_load_reads(lsg_ctx* c, const lsg_reads* r) {
    if (!c || !r) { set_error("lsg_load_reads: bad arguments"); return -2; }
    if (c->n_contigs <= 0) { set_error("lsg_load_reads: set the contigs first (the store is laid out over their tiles)"); return -2; }
    if (r->n_reads < 0 || r->n_segs < 0 || r->n_events < 0) { set_error("lsg_load_reads: negative sizes"); return -2; }
    if (r->n_segs >= 0xFFFFFFF0ll || r->n_reads >= (1ll << 29)) { set_error("lsg_load_reads: more than 2^32 segments or 2^29 reads; load in windows"); return -2; }
    if (r->n_events >= (1ll << 40)) { set_error("lsg_load_reads: more than 2^40 events"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    lsg::drop_store(c);
    const lsg_reads in = *r;                  // (r may point at c->rd's own arrays: lsg_synth_reads)
    c->rd = lsg_reads{};
    c->rd.n_reads = in.n_reads; c->rd.n_segs = in.n_segs; c->rd.n_events = in.n_events; c->rd.on_device = 1;
    const int d = in.on_device;
    if (in.n_reads > 0 && (!in.read_tid || !in.read_flag || !in.read_mapq || !in.read_cb)) { c->rd = lsg_reads{}; set_error("lsg_load_reads: NULL read array"); return -2; }
    if (in.n_segs > 0 && (!in.seg_read || !in.seg_start || !in.seg_len || !in.seg_ev_off)) { c->rd = lsg_reads{}; set_error("lsg_load_reads: NULL segment array"); return -2; }
    if (in.n_events > 0 && !in.events) { c->rd = lsg_reads{}; set_error("lsg_load_reads: NULL events array"); return -2; }
    // the small arrays (15 bytes per read, 12 per segment) are always copied: admission, the depth cap and the statistics read them.
    // Device arrays travel on a stream of their own (after whatever the caller queued on the handle's stream) while the build's first
    // kernels read the caller's arrays where they lie: a gigabyte of copies beside a kernel that waits on dependent loads.
    hipStream_t cs = nullptr;
    if (d && !c->keep_reads) {
        LSG_HIP(hipEventRecord(c->ev_copy, c->stream));
        LSG_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_copy, 0));
        cs = c->copy_stream;
    }
    if (put(c, c->b_read_tid, c->rd.read_tid, in.read_tid, in.n_reads, d, cs) ||
        put(c, c->b_read_pos, c->rd.read_pos, in.read_pos, in.n_reads, d, cs) ||
        put(c, c->b_read_flag, c->rd.read_flag, in.read_flag, in.n_reads, d, cs) ||
        put(c, c->b_read_mapq, c->rd.read_mapq, in.read_mapq, in.n_reads, d, cs) ||
        put(c, c->b_read_cb, c->rd.read_cb, in.read_cb, in.n_reads, d, cs) ||
        put(c, c->b_seg_read, c->rd.seg_read, in.seg_read, in.n_segs, d, cs) ||
        put(c, c->b_seg_start, c->rd.seg_start, in.seg_start, in.n_segs, d, cs) ||
        put(c, c->b_seg_len, c->rd.seg_len, in.seg_len, in.n_segs, d, cs)) { if (cs) (void)hipStreamSynchronize(cs); c->rd = lsg_reads{}; return -1; }
    // the events and their offsets: device arrays are read where they lie, host arrays through a staging copy; neither outlives the call
    // unless lsg_set_keep_reads asked for it
    const uint16_t* ev = in.events; const int64_t* evo = in.seg_ev_off;
    const bool staged = !d || c->keep_reads;
    if (staged && (put(c, c->b_seg_ev_off, evo, in.seg_ev_off, in.n_segs, d) || put(c, c->b_events, ev, in.events, in.n_events, d))) { c->rd = lsg_reads{}; return -1; }
    if (staged && in.events && c->hint_phased_events == (const void*)in.events) c->hint_phased_events = ev;      // (what this library's producer said about the array holds for its copy)
    const int rc = lsg::build_store(c, ev, in.n_events, evo, cs ? &in : nullptr);
    if (cs) LSG_HIP(hipStreamSynchronize(cs));                 // the handle's copies are whole before anyone counts (or the caller frees its arrays)
    if (rc) { c->rd = lsg_reads{}; lsg::drop_store(c); return rc; }      // a refused load leaves no reads behind
    if (c->keep_reads) { c->rd.events = ev; c->rd.seg_ev_off = evo; }
    else { c->b_events.release(); c->b_seg_ev_off.release(); }
    return 0;
}

int lsg_set_region(lsg_ctx* c, int32_t tid_lo, int64_t pos_lo, int32_t tid_hi, int64_t pos_hi) {
    if (!c || c->n_contigs <= 0) { set_error("lsg_set_region: set contigs first"); return -2; }
    if (tid_lo < 0 || tid_lo > c->n_contigs || tid_hi < 0 || tid_hi > c->n_contigs || (pos_lo & 63) || (pos_hi & 63) || pos_lo < 0 || pos_hi < 0) {
        set_error("lsg_set_region: bad region (positions must be multiples of 64)"); return -2;
    }
    auto tile_of = [&](int32_t tid, int64_t pos) -> uint64_t {
        if (tid >= c->n_contigs) return c->n_tiles;
        uint64_t t = (uint64_t)c->tile_base[tid] + (uint64_t)(pos >> 6);
        return t < c->tile_base[tid + 1] ? t : c->tile_base[tid + 1];
    };
    uint64_t lo = tile_of(tid_lo, pos_lo), hi = tile_of(tid_hi, pos_hi);
    if (hi < lo) { set_error("lsg_set_region: empty or inverted region"); return -2; }
    if (c->tile_lo == (uint32_t)lo && c->tile_hi == (uint32_t)hi) return 0;      // (the region that is set: a resident count stays)
    c->tile_lo = (uint32_t)lo; c->tile_hi = (uint32_t)hi;
    c->counted = c->called = false;
    return 0;
}

int lsg_pileup_count(lsg_ctx* c, const lsg_count_params* params, int64_t* n_rows, int64_t* n_columns) {
    if (!c || !params) { set_error("lsg_pileup_count: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    if (int rc = check_load_filter(c, "lsg_pileup_count", params->min_mq, params->flag_exclude, params->ignore_orphans)) return rc;
    // the load made this very count while it built the store (lsg_set_count_at_load): it is handed out once, a later call counts again
    // (a load that kept no store: its count is all there is, and stays until something invalidates it)
    const bool have = c->counted && (c->counted_at_load || c->store_skipped) && memcmp(params, &c->last_params, sizeof(*params)) == 0;
    c->counted_at_load = false;
    if (!have) { int rc = run_count(c, params); if (rc) return rc; }
    if (n_rows) for (int i = 0; i < c->n_ct; ++i) n_rows[i] = c->n_rows[i];
    if (n_columns) *n_columns = c->n_columns;
    return 0;
}

int lsg_fetch_counts(lsg_ctx* c, int32_t ct, int64_t* keys, uint8_t* ref, uint32_t* counts, int64_t capacity) {
    if (!c || !keys || !ref || !counts) { set_error("lsg_fetch_counts: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_fetch_counts(c, ct, keys, ref, counts, capacity);
}

int64_t lsg_max_live_reads(lsg_ctx* c) {
    if (!c) { set_error("lsg_max_live_reads: bad arguments"); return -1; }
    if (hipSetDevice(c->device) != hipSuccess) { set_error("lsg_max_live_reads: hipSetDevice failed"); return -1; }
    if (lsg::live_read_bound(c)) return -1;
    return c->max_live_reads;
}

int64_t lsg_max_live_reads_exact(lsg_ctx* c) {
    if (!c) { set_error("lsg_max_live_reads_exact: bad arguments"); return -1; }
    if (hipSetDevice(c->device) != hipSuccess) { set_error("lsg_max_live_reads_exact: hipSetDevice failed"); return -1; }
    if (lsg::live_read_bound_exact(c)) return -1;
    return c->max_live_exact;
}

int64_t lsg_max_live_reads_all(lsg_ctx* c) {
    if (!c) { set_error("lsg_max_live_reads_all: bad arguments"); return -1; }
    if (hipSetDevice(c->device) != hipSuccess) { set_error("lsg_max_live_reads_all: hipSetDevice failed"); return -1; }
    if (lsg::live_read_bound_all(c)) return -1;
    return c->max_live_all;
}

int lsg_get_store_shape(lsg_ctx* c, int64_t* n_entries, int64_t* n_blocks, int64_t* n_events) {
    if (!c) { set_error("lsg_get_store_shape: NULL handle"); return -2; }
    if (n_entries) *n_entries = (int64_t)c->tm_n;
    if (n_blocks) *n_blocks = (int64_t)c->tm_nblk;
    if (n_events) *n_events = c->tm_events;
    return 0;
}

int lsg_get_build_times(lsg_ctx* c, float* ms4) {
    if (!c || !ms4) { set_error("lsg_get_build_times: bad arguments"); return -2; }
    for (int i = 0; i < 4; ++i) ms4[i] = c->build_ms[i];
    return 0;
}

int lsg_get_count_stats(lsg_ctx* c, lsg_count_stats* out) {
    if (!c || !out) { set_error("lsg_get_count_stats: bad arguments"); return -2; }
    *out = c->stats;
    return 0;
}

int lsg_get_layout_info(lsg_ctx* c, int32_t* path, double* build_ms, int64_t* store_bytes) {
    if (!c) { set_error("lsg_get_layout_info: NULL handle"); return -2; }
    if (path) *path = c->store_skipped ? (c->wsh ? 6 : c->line_loads ? 5 : 4) : c->load_was_fused ? 3 : 2;       // 3: the load's gather also made the first count (lsg_set_count_at_load)
    if (build_ms) *build_ms = c->layout_build_ms;
    if (store_bytes) {
        int64_t b = 0;
        for (auto& x : c->tm) b += (int64_t)x.cap;
        for (auto& x : c->bt) b += (int64_t)x.cap;
        for (DevBuf* x : {&c->d_tile_cap, &c->d_tile_off, &c->b_events, &c->b_seg_ev_off, &c->b_read_tid, &c->b_read_pos, &c->b_read_flag, &c->b_read_mapq, &c->b_read_cb,
                          &c->b_seg_read, &c->b_seg_start, &c->b_seg_len}) b += (int64_t)x->cap;
        b += (int64_t)c->ws[WS_SEG_INFO].cap;
        *store_bytes = b;
    }
    return 0;
}

int lsg_call_step1(lsg_ctx* c, const lsg_call_params* params, int64_t* n_sites, int64_t* n_candidates) {
    if (!c || !params) { set_error("lsg_call_step1: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    int rc = run_call(c, params);
    if (rc) return rc;
    if (n_sites) *n_sites = c->n_sites;
    if (n_candidates) *n_candidates = c->n_cand;
    return 0;
}

int lsg_fetch_calls(lsg_ctx* c, lsg_call* out, int64_t capacity, int32_t candidates_only, int64_t* n_out) {
    if (!c || (!out && capacity > 0)) { set_error("lsg_fetch_calls: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_fetch_calls(c, out, capacity, candidates_only, n_out);
}

int lsg_export_calls(lsg_ctx* c, int32_t kind, void* dst_device, int64_t capacity, int64_t* n_out) {
    if (!c || kind < 0 || kind > 2) { set_error("lsg_export_calls: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_select_calls(c, kind, reinterpret_cast<lsg_call*>(dst_device), capacity, n_out);
}

int lsg_set_table_names(lsg_ctx* c, int32_t n_contigs, const char* contig_names, int32_t n_celltypes, const char* celltype_names) {
    if (!c || n_contigs < 0 || !contig_names || !celltype_names) { set_error("lsg_set_table_names: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_set_table_names(c, n_contigs, contig_names, n_celltypes, celltype_names);
}
int lsg_format_table(lsg_ctx* c, int32_t table, int64_t* n_bytes) {
    if (!c) { set_error("lsg_format_table: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_format_table(c, table, n_bytes);
}
int lsg_copy_table(lsg_ctx* c, int32_t table, char* dst_host, int64_t capacity) {
    if (!c || (!dst_host && capacity > 0) || capacity < 0) { set_error("lsg_copy_table: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_copy_table(c, table, dst_host, capacity);
}
int lsg_append_table(lsg_ctx* c, int32_t table, const char* path) {
    if (!c || !path || !*path) { set_error("lsg_append_table: bad arguments"); return -2; }
    return run_append_table(c, table, path);
}
int lsg_step2_summary(lsg_ctx* c, int32_t n_cols, uint8_t* kinds, int64_t* n_survivor_bytes) {
    if (!c || !kinds) { set_error("lsg_step2_summary: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_step2_summary(c, n_cols, kinds, n_survivor_bytes);
}
int lsg_free_table(lsg_ctx* c, int32_t table) {
    if (!c) { set_error("lsg_free_table: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_free_table(c, table);
}

int lsg_load_posset(lsg_ctx* c, int32_t kind, const int64_t* keys, int64_t n, int32_t on_device) {
    if (!c || kind < 0 || kind >= 3 || n < 0 || (n > 0 && !keys)) { set_error("lsg_load_posset: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    PosSet& s = c->posset[kind];
    s.n = 0;
    if (n == 0) return 0;
    if (s.keys.reserve((size_t)n * 8)) return -1;
    LSG_HIP(hipMemcpyAsync(s.keys.p, keys, (size_t)n * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    LSG_HIP(hipStreamSynchronize(c->stream));
    s.n = n;
    return 0;
}

int lsg_probe_posset(lsg_ctx* c, int32_t kind, const int64_t* keys, int64_t n, uint8_t* hits, int32_t on_device) {
    if (!c || kind < 0 || kind >= 3 || n < 0 || (n > 0 && (!keys || !hits))) { set_error("lsg_probe_posset: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_probe(c, kind, keys, n, hits, on_device);
}

int lsg_genotype_cells(lsg_ctx* c, const lsg_genotype_params* params, int64_t n_sites, const int64_t* site_keys,
                       const uint8_t* alt_sym, uint32_t* dp, uint32_t* alt, int32_t on_device) {
    if (!c || !params) { set_error("lsg_genotype_cells: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    if (int rc = check_load_filter(c, "lsg_genotype_cells", params->min_mq, params->flag_exclude, params->ignore_orphans)) return rc;
    return run_genotype(c, params, n_sites, site_keys, alt_sym, dp, alt, on_device, 0, 0, nullptr);
}

int lsg_genotype_cells_grouped(lsg_ctx* c, const lsg_genotype_params* params, int32_t max_depth, int64_t n_sites, const int64_t* site_keys, const uint8_t* alt_sym,
                               int64_t n_groups, const int64_t* group_off, uint32_t* dp, uint32_t* alt, int32_t on_device) {
    if (!c || !params || n_groups < 0 || (n_groups > 0 && !group_off)) { set_error("lsg_genotype_cells_grouped: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    if (int rc = check_load_filter(c, "lsg_genotype_cells_grouped", params->min_mq, params->flag_exclude, params->ignore_orphans)) return rc;
    if (n_groups > 0 && (group_off[0] != 0 || group_off[n_groups] != n_sites)) { set_error("lsg_genotype_cells_grouped: the groups must cover the sites"); return -2; }
    return run_genotype(c, params, n_sites, site_keys, alt_sym, dp, alt, on_device, max_depth, n_groups, group_off);
}

int lsg_betabinom_sf4(lsg_ctx* c, int64_t n_items, const uint32_t* k, const uint32_t* n, double alpha, double beta, int32_t* out_p4) {
    if (!c) { set_error("lsg_betabinom_sf4: NULL handle"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_sf4(c, n_items, k, n, alpha, beta, out_p4, nullptr);
}

int lsg_betabinom_sf(lsg_ctx* c, int64_t n_items, const uint32_t* k, const uint32_t* n, double alpha, double beta, int32_t* out_p4, double* out_p) {
    if (!c || !out_p) { set_error("lsg_betabinom_sf: bad arguments"); return -2; }
    LSG_HIP(hipSetDevice(c->device));
    return run_sf4(c, n_items, k, n, alpha, beta, out_p4, out_p);
}

} // extern "C"
