// Internal state of one liblongsom_hip handle (one per GPU).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/longsom_hip.h"
#include "../../include/longsom_synth.h"

namespace lsg {

constexpr int TILE_W = 64;          // reference positions per tile = one lane per position
constexpr int ROW_PLANES = 34;      // count-row planes kept in HBM: DP, NC, CC[8], BC[8], BQ[8], BCf[8]; BCr = BC - BCf is derived on export
constexpr int NCTR = 33;            // accumulators kept per position: NC, CC[8], BC[8], BQ[8], BCf[8]

// Resident count rows of one cell type: blocks of 64 rows, [block][quad][64 rows][4 planes] with quad = plane / 4 (34 planes in
// 9 quads, the last two words unused).  All words of a row lie inside one 9216-byte block (a unit's rows, <= 64 and consecutive,
// inside two), so a unit is written and read around ONE place in memory, and a lane moves four planes of its row with one 16-byte
// access: 9 stores per unit instead of 34, 4 loads per row for the call stage instead of 14.  row_cap is a multiple of 64.
constexpr int ROW_QUADS = (ROW_PLANES + 3) / 4;
constexpr uint64_t ROW_BLOCK_WORDS = (uint64_t)ROW_QUADS * 256;
constexpr uint64_t ROW_STORED_WORDS = (uint64_t)ROW_QUADS * 4;          // words of HBM per row (36)
constexpr uint32_t ROW_NARROW = 1u << 31;     // in a unit's row base: its rows are 16-bit planes in the first half of their blocks (emit_unit)
__host__ __device__ inline uint64_t row_word(uint64_t row, int plane) {
    return (row >> 6) * ROW_BLOCK_WORDS + (uint64_t)(plane >> 2) * 256 + (row & 63) * 4 + (uint64_t)(plane & 3);
}

void set_error(const char* fmt, ...);
const char* get_error();

#define LSG_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            lsg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t slack = bytes / 16 < ((size_t)64 << 20) ? bytes / 16 : ((size_t)64 << 20);      // (a 124 GB buffer must not ask for 8 more)
        size_t want = bytes + slack + 256;
        LSG_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PosSet { DevBuf keys; int64_t n = 0; };

// workspace buffers (lsg_ctx::ws)
enum { WS_NE_NSLOT = 0, WS_NE_ACC, WS_NE_GEOM, WS_MULTI_LIST, WS_MACC, WS_EXPORT_K, WS_EXPORT_R,
       WS_EXPORT_C, WS_CALL_FLAGS, WS_CALL_SEL, WS_CALL_CANDS, WS_CALL_TASKS, WS_SEG_INFO };

// ---- tile store (store.hip builds it at load, pileup.hip counts over it, genotype.hip looks single sites up in it) -------------
// The resident form of the reads' events: every (segment x 64-position tile) ENTRY of a read that carries a barcode, the entries of a
// tile adjacent and sorted by barcode, eight entries to a 1 KB block held transposed ([position 0..63][entry 0..7], 16 bytes per
// position), every tile padded to whole blocks.  Independent of count parameters and of the barcode -> cell-type table.
//   s0[p]  cb [0..23] | upper pileup window << 29 | forward << 30 | first entry of its barcode's run in the tile << 31   (pad entries: cb = CB_MASK, run start)
//          An entry never crosses an edge of the reference's pileup windows ([1, 50001), [50001, ...: BaseCellCounter.py:81-113,185-191): in
//          the one tile per window edge a segment makes two entries, and bit 29 says the entry lies in the window that starts inside its
//          tile (what the per-window max_depth rule needs: layout.hip depth_cap_drops)
//   b[p]   events - 1 [0..5] | first entry of its segment << 6 | run of exactly one entry << 7
//   rd[p]  owning read: admission under a count's read filters (SAM flag, MAPQ) and the pileup's max_depth rule are decided per READ
//          (a bit per read, made per count: pileup.hip k_read_admit) and looked up through it
constexpr uint32_t CB_MASK = 0x00FFFFFFu;
constexpr uint32_t TM_RUNSTART = 1u << 31, TM_FWD = 1u << 30, TM_WHI = 1u << 29;
constexpr uint32_t TM_PAD_S0 = CB_MASK | TM_RUNSTART;
constexpr int TM_GROUP = 4;                 // blocks a wave of the walk loads per group: the arrays are padded by one group
enum { TM_STORE = 0, TM_S0, TM_B, TM_RD, TM_META, TM_BLK_TILE, TM_BLK_OFF, TM_EXT,                                // per load
       TM_JOBS, TM_NE_UNITS, TM_NE_GEOM, TM_NE_NSLOT, TM_NE_ACC, TM_MULTI, TM_CHUNKS, TM_NBUF };                          // the plan: per load and number of cell types
// one job of the walk: a tile, or a run-aligned piece of a deep one.  e0, e1: padded-entry range; w0: unit of (tile, cell type 0);
// slab: of (job, cell type 0) or ~0; nj: jobs of the tile, bit 31 = the job is longer than the packed planes' fields hold (k_tm_walk_wide
// takes it); cnt: entries of the tile; emid: where the job's second wave starts (a run start, or e1)
// base, off: the tile's first padded entry (8 x blocks before it) and its first entry in the sort's output; tstart, tid: the tile's
// first position and contig - everything a job's workgroup needs in ONE record (k_tm_gather_count fetches the next job's while it works)
struct TmJob { uint32_t e0, e1, w0, slab, nj, cnt, tile, emid, base, off; int32_t tstart, tid; };
constexpr int TM_JOB_WORDS = 12;
static_assert(sizeof(TmJob) == 4 * TM_JOB_WORDS, "TmJob layout");
constexpr uint32_t TMJ_WIDE = 1u << 31;
constexpr int TM_JOB_TGT = 3072, TM_JOB_LIMIT = 4095;      // LIMIT: what the planes' 12-bit forward field holds; a cut moves forward to the next run start

} // namespace lsg

struct lsg_ctx;
namespace lsg {
int build_store(lsg_ctx* c, const uint16_t* events, int64_t n_events, const int64_t* seg_ev_off, const lsg_reads* src = nullptr);   // store.hip: the load's tile store from the caller's compact events (src: read the per-read / per-segment arrays there instead of the handle's copies, which may still be under way)
int ensure_plan(lsg_ctx* c);       // store.hip: jobs / units / slabs of a count over the store for the current number of cell types
void drop_store(lsg_ctx* c);       // store.hip: new reads or contigs
int live_read_bound(lsg_ctx* c);   // layout.hip: fills max_live_reads when it is stale (-1)
int live_read_bound_exact(lsg_ctx* c);   // layout.hip: fills max_live_exact (per cell type, per position: the largest buffer a push can meet, counting the pushed read)
int live_read_bound_all(lsg_ctx* c);
__global__ void k_read_end(const uint32_t* seg_read, const int32_t* seg_start, const int32_t* seg_len, int64_t n_segs, int32_t* read_end);      // layout.hip
__global__ void k_read_end_init(const int32_t* read_pos, int64_t n_reads, int32_t* read_end);
int depth_cap_drops(lsg_ctx* c, const lsg_count_params* p);   // layout.hip: htslib's max_depth rule -> d_read_drop (or none)
struct GatherCountSrc { const uint16_t* events; int64_t n_events; const uint64_t* key; const uint32_t* rdv; int32_t cb_bits; int32_t src_shift; int32_t wsh; };
int run_gather_count(lsg_ctx* c, const lsg_count_params* p, const GatherCountSrc& src, bool direct);      // pileup.hip: the load's gather and the first count in one pass (k_tm_gather_count), or the count alone from the caller's events (k_tm_count_direct)
}

struct lsg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;      // the load's copies of the caller's device arrays run here, beside the build's first kernels
    hipEvent_t ev_copy = nullptr;
    hipEvent_t ev_blk = nullptr;            // store.hip: the blocks' tiles are made on the copy stream
    hipEvent_t ev_lpt = nullptr;            // ... and the sort's segments ordered deepest first
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // store.hip: the build's phases

    // genome
    int32_t n_contigs = 0;
    std::vector<int64_t> contig_len;
    std::vector<uint32_t> tile_base;      // [n_contigs+1] first tile of each contig
    uint32_t n_tiles = 0;
    uint32_t tile_lo = 0, tile_hi = 0;    // counted tile range (lsg_set_region); default = everything
    lsg::DevBuf d_tile_base, d_contig_len, d_ref_ptrs;
    std::vector<lsg::DevBuf> ref;         // per contig, owned copies
    std::vector<const uint8_t*> ref_ptr;  // per contig device pointer (owned or adopted)

    // barcodes
    int32_t n_cb = 0, n_ct = 0;
    lsg::DevBuf d_celltype_of, d_ct_rank;   // ct_rank[cb] = rank of the barcode within its cell type
    std::vector<uint8_t> h_celltype_of;     // the table as it was set (lsg_set_barcodes leaves an identical one alone)
    uint32_t ct_size[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};

    // reads
    lsg_reads rd{};                       // device pointers
    lsg::DevBuf b_read_tid, b_read_pos, b_read_flag, b_read_mapq, b_read_cb;
    lsg::DevBuf b_seg_read, b_seg_start, b_seg_len, b_seg_ev_off, b_events;
    int32_t plp_window = 50000, st_window = 50000;        // lsg_set_pileup_window: the reference's pileup windows (--bin) for the next loads / of the resident store
    int32_t lf_min_mq = 0, lf_ignore_orphans = 0; uint32_t lf_flag_exclude = 0;      // lsg_set_load_filter: reads failing it are not stored by the next loads
    int32_t st_min_mq = 0, st_ignore_orphans = 0; uint32_t st_flag_exclude = 0;      // ... and the filter the resident store was built under
    bool keep_unlisted = false;           // lsg_set_keep_unlisted: the BAM loads keep reads without a listed barcode (cb = -1; never counted, but in the genotyping pileup's buffer)
    bool keep_reads = false;              // lsg_set_keep_reads: the compact events stay resident beside the store (rd.events; tests, sampling)
    // tile store (see above) and the plan of a count over it
    lsg::DevBuf d_tile_cap, d_tile_off;   // entries per tile and their exclusive prefix
    lsg::DevBuf tm[lsg::TM_NBUF];
    lsg::DevBuf bt[18];                   // temporaries of the build (kept while they are small against the device: allocation is what a rebuild would wait for)
    uint64_t tm_n = 0;                    // entries
    int64_t tm_events = 0;                // events they hold
    uint64_t tm_np = 0;                   // padded entries = 8 x blocks
    uint32_t tm_nblk = 0, tm_njobs = 0, tm_nchunks = 0, tm_n_ne = 0, tm_n_multi = 0, tm_n_slabs = 0, tm_n_wide = 0;
    int plan_n_ct = 0;                    // cell types the plan was made for (0: none)
    int plan1_n_ct = 0;                   // ... and its tile-level half (units, jobs, slabs per tile and their totals), which the load can make beside its gather
    uint32_t plan1_tot[4] = {0, 0, 0, 0};
    uint32_t plan_misc[2] = {0, 0};       // wide jobs, chunks (read back from the job-level half)
    uint32_t* d_plan_misc = nullptr;      // ... where they lie on the device
    // lsg_set_count_at_load: the next load also makes the first count under these parameters, in the pass that builds the store
    bool cal_enabled = false; lsg_count_params cal_params{};
    bool counted_at_load = false;         // the resident count is that one and nobody has asked for it yet
    bool load_was_fused = false;          // the last load built its store in the pass that counted (lsg_get_layout_info path 3)
    bool tm_valid = false;
    int32_t store_policy = 0;             // lsg_set_store_policy: LSG_STORE_KEEP / LSG_STORE_SKIP_WHEN_COUNTED
    bool store_skipped = false;           // the last load made its count and kept no store: what needs one fails
    double layout_build_ms = 0;           // wall time of the last build (lsg_get_layout_info)
    float build_ms[4] = {0, 0, 0, 0};     // HIP-event times of the last build: capacities + scatter, sort, fill, gather
    int64_t max_live_reads = -1;          // layout.hip: bound on the reads live at once in the reference's pileup buffer (-1 = stale)
    int64_t max_live_all = -1;            // the same over all reads with a barcode: table-independent, cached per load
    int32_t wsh = 0;                      // build_store: the bins of the last load's entries - 0: the 64-position tiles, 1: 128-position windows (a load that kept no store)
    bool win_off = false;                 // ... set while a load that could not be made by windows after all is made again by tiles
    int32_t events_layout = 0;            // lsg_set_events_layout: what the caller says about the events of the next loads (LSG_LAYOUT_*)
    const void* hint_phased_events = nullptr;      // the events array this library's own producer (synth.hip, ingest.hip) last laid out phased modulo 128
    bool src_phased = false;              // build_store: the last load's events were tile-phased (LSG_LAYOUT_PHASED): an entry = one 128-byte line
    bool line_loads = false;              // pileup.hip run_gather_count: the last direct count fetched every entry as its one 128-byte line (tile-phased events)
    bool keys_only_off = false;           // build_store: this load sorts values with its keys (set while a load of keys alone is made again)
    int64_t max_live_exact = -1;          // per cell type at position resolution (asked only when the tile-level bounds cannot rule the cap out)
    lsg::DevBuf d_read_drop;              // layout.hip: per read, the pileup's max_depth rule under the last count's parameters: 1 = dropped in every window it overlaps, 2 = in some (d_drop_pairs)
    lsg::DevBuf d_drop_pairs; int64_t n_drop_pairs = 0;      // sorted (read << 32 | window of its contig) of the reads dropped in some windows only
    lsg::DevBuf d_read_adm;               // pileup.hip: a bit per read, admitted under the current count's read filters (made only when some stored read can fail them)
    bool has_drops = false;
    int64_t n_depth_dropped = 0;

    // count-stage workspace
    lsg::DevBuf d_read_key;
    lsg::DevBuf d_ne_units, d_ne_mask, d_ne_rowbase, d_ne_rowoff, d_scalars, d_cub_tmp, d_ix_stat;
    lsg::DevBuf d_rows[LSG_MAX_CELLTYPES]; // blocked planes, see lsg::row_word
    uint64_t row_cap = 0;
    uint32_t arena = 256;                  // rows a wave reserves per allocation in the current count (multiple of 256)
    uint32_t n_ne = 0;
    int64_t n_rows[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};
    int64_t n_columns = 0;
    lsg_count_params last_params{};
    lsg_count_stats stats{};
    bool counted = false;

    // call stage
    lsg::DevBuf d_calls, d_site_off, d_tail_table;      // d_tail_table: call.hip, memo of the small-n beta-binomial tails
    double tail_table_key[4] = {0, 0, 0, 0}; bool tail_table_valid = false;
    int64_t n_sites = 0, n_cand = 0, n_pass = -1;      // n_pass: PASS candidates listed by k_call_finish (-1: no list)
    lsg::DevBuf d_pass_list, d_defer_list, d_xcd_queues;
    uint32_t call_tasks_per_site = 1;     // call.hip: how the tail tasks' buffers are sized (grown on demand)
    bool called = false;

    // the tables' text (tables.hip): flat copies of the count rows, the names the rows print, one device buffer per table
    lsg::DevBuf tab_keys[LSG_MAX_CELLTYPES], tab_refs[LSG_MAX_CELLTYPES], tab_rows[LSG_MAX_CELLTYPES], tab_names, tab_scratch, tab_scratch2;
    int32_t tab_scratch_table = -1; int64_t tab_scratch_rows = 0;      // whose rows' lengths and places tab_scratch holds
    uint64_t tab_rows_serial[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};      // count_serial the flat copy was made of (0 = none)
    uint64_t count_serial = 0;            // bumped by every count / lsg_load_counts
    lsg::DevBuf tab_text[LSG_TABLE_SLOTS];
    int64_t tab_bytes[LSG_TABLE_SLOTS] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};      // -1: not formatted
    int32_t tab_n_contigs = 0, tab_n_ct = 0; uint32_t tab_ct_off_at = 0, tab_order_at = 0, tab_ct_txt_at = 0, tab_contig_txt_at = 0;

    lsg::PosSet posset[3];
    lsg::DevBuf syn[12];                  // synthetic-model tables + scan scratch (synth.hip)
    lsg::DevBuf gen[10];                  // the arrays lsg_synth_generate last produced
    lsg::DevBuf ws[16];                   // count-stage workspace (pileup.hip, enum WS_*)
    int n_cus = 256;
    unsigned long long* h_pin = nullptr;  // 4 KB of pinned host memory: landing zone of the small device -> host reads between phases
    uint32_t n_multi = 0;
};
