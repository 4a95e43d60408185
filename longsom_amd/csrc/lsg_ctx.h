// Internal state of one liblongsom_hip handle (one per GPU).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/longsom_hip.h"

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
        size_t want = bytes + bytes / 16 + 256;
        LSG_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PosSet { DevBuf keys; int64_t n = 0; };

// workspace buffers (lsg_ctx::ws)
enum { WS_SLOT_PEX = 0, WS_NE_NSLOT, WS_NE_SLOT_BASE, WS_NE_ACC, WS_NE_GEOM, WS_SLOT_W, WS_SLOT_CNT,
       WS_SLOT_OFF, WS_ENT, WS_CHUNK_START, WS_REC, WS_SLOT_LIST, WS_MULTI_LIST, WS_MACC, WS_EXPORT_K, WS_EXPORT_R,
       WS_EXPORT_C, WS_SLICES, WS_HUGE_LIST, WS_CALL_FLAGS, WS_CALL_SEL, WS_CALL_CANDS, WS_CALL_TASKS, WS_SEG_INFO };


} // namespace lsg

struct lsg_ctx;
namespace lsg {
int relayout_events(lsg_ctx* c);   // layout.hip: tile-aligned copy of the resident events
int live_read_bound(lsg_ctx* c);   // layout.hip: fills max_live_reads when it is stale (-1)
int depth_cap_drops(lsg_ctx* c, const lsg_count_params* p);   // layout.hip: htslib's max_depth rule -> d_read_drop (or none)
}

struct lsg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

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
    uint32_t ct_size[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};

    // reads
    lsg_reads rd{};                       // device pointers
    lsg::DevBuf b_read_tid, b_read_pos, b_read_flag, b_read_mapq, b_read_cb;
    lsg::DevBuf b_seg_read, b_seg_start, b_seg_len, b_seg_ev_off, b_events;
    uint64_t entries_upper = 0;           // sum over segments of tiles overlapped
    // static per load: entries a tile can ever hold (segments of reads with a barcode that touch it) and their exclusive prefix = the
    // tile's region of the entry buffer; with <= 2 cell types the scatter fills a region from both ends and needs no counting pass
    lsg::DevBuf d_tile_cap, d_tile_off, d_cur_lo, d_cur_hi;
    bool tile_caps_valid = false;
    // pileup.hip "tile index": every tile's entries sorted by barcode ONCE per load (independent of parameters and of the barcode ->
    // cell-type table); a count then resolves admission and cell type in one streaming pass instead of scattering and sorting again
    lsg::DevBuf d_ix0, d_ix1, d_ix2, d_ix_netile, d_ix_chunk, d_ix_carry, d_ix_stat;
    uint64_t ix_n = 0;                    // static entries
    uint32_t ix_n_netile = 0;             // tiles that hold any
    bool index_valid = false;
    bool index_path = false;              // the last / current count runs on the tile index
    // pileup.hip "tile-major store": the admitted entries' events in index order, eight entries to a transposed 1 KB block, with the
    // static job / unit / slab tables of a count over it; keyed on the read filters and the number of cell types
    lsg::DevBuf tm[16];
    lsg::DevBuf bt[10];                   // temporaries of the index / store build (kept: device allocation is what a rebuild would wait for)
    uint64_t tm_np = 0;                   // padded entries
    uint32_t tm_nblk = 0, tm_njobs = 0, tm_nchunks = 0, tm_n_ne = 0, tm_n_multi = 0, tm_n_slabs = 0;
    int64_t tm_key[4] = {0, 0, 0, 0};     // min_mq, flag_exclude, ignore_orphans, n_ct
    bool tm_valid = false, tm_usable = false;
    bool tm_path = false;                 // the last / current count runs on the tile-major store
    double layout_build_ms = 0;           // wall time spent building the index / store for the current reads (lsg_get_layout_info)
    int layout_policy = 0;                // lsg_set_layout_policy: 0 auto (from the second count of a load), 1 eager, 2 never
    int64_t seen_key[4] = {-1, -1, -1, -1}; uint32_t seen_counts = 0;      // read filters of the last count and how many counts of this load used them
    int64_t max_live_reads = -1;          // layout.hip: bound on the reads live at once in the reference's pileup buffer (-1 = stale)
    int64_t max_live_all = -1;            // the same over all reads with a barcode: table-independent, cached per load
    lsg::DevBuf d_read_drop;              // layout.hip: per read, 1 = dropped by the pileup's max_depth rule under the last count's parameters
    bool has_drops = false;
    int64_t n_depth_dropped = 0;

    // count-stage workspace
    lsg::DevBuf d_read_key, d_unit_cnt, d_unit_off, d_unit_fill;
    lsg::DevBuf d_ne_units, d_ne_mask, d_ne_rowbase, d_ne_rowoff, d_scalars, d_cub_tmp;
    lsg::DevBuf d_rows[LSG_MAX_CELLTYPES]; // blocked planes, see lsg::row_word
    uint64_t row_cap = 0;
    uint32_t arena = 256;                  // rows a wave reserves per allocation in the current count (multiple of 256)
    uint32_t n_ne = 0, n_deep = 0;
    int64_t n_rows[LSG_MAX_CELLTYPES] = {0, 0, 0, 0};
    int64_t n_columns = 0;
    lsg_count_params last_params{};
    lsg_count_stats stats{};
    bool counted = false;

    // call stage
    lsg::DevBuf d_calls, d_site_off, d_tail_table;      // d_tail_table: call.hip, memo of the small-n beta-binomial tails
    double tail_table_key[4] = {0, 0, 0, 0}; bool tail_table_valid = false;
    int64_t n_sites = 0, n_cand = 0, n_pass = -1;      // n_pass: PASS candidates listed by k_call_finish (-1: no list)
    lsg::DevBuf d_pass_list;
    bool called = false;

    lsg::PosSet posset[3];
    lsg::DevBuf syn[12];                  // synthetic-model tables + scan scratch (synth.hip)
    lsg::DevBuf ws[32];                   // count-stage workspace (pileup.hip, enum WS_*)
    int n_cus = 256;
    unsigned long long* h_pin = nullptr;  // 4 KB of pinned host memory: landing zone of the small device -> host reads between phases
    uint32_t n_slots = 0, n_multi = 0;
};
