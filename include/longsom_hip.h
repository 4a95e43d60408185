/*
 * longsom_hip.h — C-ABI of liblongsom_hip.so, the MI355X (gfx950) implementation of
 * LongSom's SComatic-derived SNV hot path:
 *
 *   SplitBamCellTypes -> BaseCellCounter -> MergeBaseCellCounts -> BaseCellCalling.step1
 *
 * The reference has no FFI for this path (it is pure Python): the operator
 * boundary is the Snakemake rule contract (files in / files out).  This header is the boundary
 * a maintainer binds with ctypes from the rule scripts; each entry point
 * names the reference code it replaces (paths relative to /root/reference/workflow/scripts).
 *
 * Conventions
 *   - plain C, pointers + sizes only; no torch / C++ types.
 *   - every call returns 0 on success, <0 on error; lsg_last_error() gives the message
 *     (thread-local).  One handle per GPU; a handle is not thread-safe; handles are independent.
 *   - "on_device" != 0 means the array pointers are device pointers on the handle's GPU
 *     (e.g. torch tensors' data_ptr()); 0 means host memory; either way borrowed for the duration of the call
 *     (exception: lsg_load_reference adopts a device array).
 *   - positions are 0-based on the device; text writers add 1 (BaseCellCounter.py:288).
 *   - there is NO CPU fallback in this library: without a HIP device lsg_create() fails.
 */
#ifndef LONGSOM_HIP_H
#define LONGSOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsg_ctx lsg_ctx;

/* Symbol classes of one pileup entry, in the allele order of the reference
 * (BaseCellCalling.step1.py:20  Alleles = ["A","C","T","G","I","D","N","O"]);
 * LSG_SYM_NA = the reference's 'NA' (intron '>' '<', IUPAC, '=': EasyReadPileup,
 * BaseCellCounter.py:152-180) and is never counted. */
enum { LSG_SYM_A = 0, LSG_SYM_C = 1, LSG_SYM_T = 2, LSG_SYM_G = 3, LSG_SYM_I = 4,
       LSG_SYM_D = 5, LSG_SYM_N = 6, LSG_SYM_O = 7, LSG_SYM_NA = 15 };

/* One count row = 42 uint32 per (site, cell type):
 *   [0] DP   [1] NC   [2..9] CC   [10..17] BC   [18..25] BQ   [26..33] BCf   [34..41] BCr
 * each vector indexed by the symbol class above (the TSV prints classes 0..5 only,
 * BaseCellCounter.py:300-306). */
#define LSG_EVENT_VALID 0x0800u
#define LSG_EVENT(sym, qual) ((uint16_t)((sym) < 8 ? (LSG_EVENT_VALID | ((sym) << 8) | ((qual) & 0xffu)) : 0u))
#define LSG_EVENT_SYM(ev) (((ev) & LSG_EVENT_VALID) ? (((ev) >> 8) & 7u) : (unsigned)LSG_SYM_NA)
#define LSG_EVENT_QUAL(ev) ((ev) & 0xffu)
#define LSG_ROW_WORDS 42
#define LSG_MAX_CELLTYPES 4

/* Pre-decoded read-record arrays ("SoA" form of a coordinate-sorted BAM).
 * A read is split into SEGMENTS = maximal runs of consecutive reference positions that carry a
 * pileup entry (M/=/X/D columns; N reference skips break segments).  One EVENT per covered
 * reference position: uint16 = LSG_EVENT(symbol_class, base_quality) = 0x0800 | class << 8 | quality for the
 * eight countable classes and 0 for LSG_SYM_NA (so that "no event" and "not countable" are both 0: the
 * kernels read lanes outside an entry's range as 0), where the symbol class and the
 * quality follow htslib bam_plp + pysam PileupColumn semantics as used by BaseCellCounter.py:191-216
 * (anchor base of an indel -> I / D, interior deletion column -> O with the quality of the next
 * query base). */
/* Where a segment's events lie in `events` is the producer's choice (seg_ev_off; n_events = the array's extent, gaps are never read).
 * Two layouts are made by this library's own producers (the device BAM decoder, the synthetic generators):
 *   LSG_LAYOUT_COMPACT  segment after segment, no gaps;
 *   LSG_LAYOUT_PHASED   "tile-phased": every read's region starts at a multiple of 128 events and every segment at an offset congruent
 *                       to its reference start modulo 128 (gaps hold 0), so that the events a read has inside one 128-position window of the
 *                       pileup (tiles 2 w and 2 w + 1) lie inside ONE aligned 256-byte block of the array, and those inside one 64-position
 *                       tile inside one aligned 128-byte line: the count fetches every block once, with one load instruction.
 * lsg_load_reads recognises a phased array by looking (every admitted segment's (seg_ev_off - seg_start) is a multiple of 64 / 128, `events`
 * is 128 / 256-byte aligned).  Whether it bins a load by windows or by tiles has to be decided before it has looked, though: a caller whose
 * arrays are phased modulo 128 says so with lsg_set_events_layout (this library's own producers do); a load that finds the claim wrong
 * starts again by tiles.  A caller's own arrays may use either layout, or anything else. */
enum { LSG_LAYOUT_COMPACT = 0, LSG_LAYOUT_PHASED = 1 };
typedef struct {
    int64_t n_reads;
    int64_t n_segs;
    int64_t n_events;
    /* per read */
    const int32_t*  read_tid;    /* contig index                                  */
    const int32_t*  read_pos;    /* 0-based leftmost position                     */
    const uint16_t* read_flag;   /* SAM flag                                      */
    const uint8_t*  read_mapq;   /* MAPQ                                          */
    const int32_t*  read_cb;     /* dense barcode id, -1 = no CB tag / not in barcodes.tsv */
    /* per segment */
    const uint32_t* seg_read;    /* owning read index                             */
    const int32_t*  seg_start;   /* 0-based reference start                       */
    const int32_t*  seg_len;     /* number of reference positions (= events)      */
    const int64_t*  seg_ev_off;  /* index of the segment's first event            */
    /* per event */
    const uint16_t* events;
    int32_t on_device;
} lsg_reads;

/* Parameters of the count stage; defaults = the flags LongSom's rules pass
 * (R:SNVCalling.smk:52-59 + script defaults BaseCellCounter.py:331-339). */
typedef struct {
    int32_t min_bq;        /* --min_bq  20 */
    int32_t min_mq;        /* --min_mq  60 (config.yaml:74) */
    int32_t min_dp;        /* --min_dp  5  */
    int32_t min_cc;        /* --min_cc  5  */
    uint32_t flag_exclude; /* reads with any of these SAM flag bits are dropped:
                              0x4|0x100|0x200|0x400 (pysam pileup flag_filter) | 0x800
                              (BaseCellCounter.py:249) = 0xF04 */
    int32_t ignore_orphans;/* 1: drop paired reads that are not proper pairs (pysam default) */
    int32_t max_depth;     /* bam.pileup(..., max_depth = 200000), BaseCellCounter.py:191 — htslib's bam_plp_push rule, applied to
                              every cell type's read stream (= its SplitBam output): a read that is not the first of its start
                              position is dropped while the reads still buffered (end >= that position, counting secondary-free,
                              MAPQ- and orphan-filtered reads INCLUDING supplementary ones) + 1 exceed max_depth.  0 = no cap.
                              Costs nothing while lsg_max_live_reads() <= max_depth (no read can be dropped then). */
} lsg_count_params;

/* Parameters of the step-1 call (BaseCellCalling.step1.py:585-604). */
typedef struct {
    double alpha1, beta1, alpha2, beta2;
    int32_t min_cov;        /* --min_cov 5        */
    int32_t min_cells;      /* --min_cells 5      */
    int32_t min_ac_cells;   /* --min_ac_cells 2   */
    int32_t min_ac_reads;   /* --min_ac_reads 3   */
    int32_t max_cell_types; /* --max_cell_types 1 */
    int32_t min_cell_types; /* --min_cell_types 2 */
} lsg_call_params;

/* One step-1 call record per merged site (fixed size; the host formats the TSV line).
 * p-values are stored as the integer k with p = k / 10000 after Python round(x, 4)
 * (round-half-even on the exact binary value). */
#define LSG_CALL_MAX_ALT 4   /* 4 when REF is not one of A,C,G,T (IUPAC reference base) */
typedef struct {
    int64_t  key;                       /* (tid << 32) | pos0                                   */
    uint8_t  ref;                       /* reference base (ASCII, upper)                        */
    uint8_t  present;                   /* bit c: cell type c has a count row at this site      */
    uint8_t  considered;                /* bit c: cell type c passed min_cov / min_cells         */
    uint8_t  has_cand;                  /* bit c: cell type c has >=1 alt candidate             */
    uint8_t  n_alt[LSG_MAX_CELLTYPES];  /* candidates per cell type                             */
    uint8_t  alt[LSG_MAX_CELLTYPES][LSG_CALL_MAX_ALT];    /* symbol class, sorted as 'A'<'C'<'G'<'T' */
    uint8_t  ct_filter[LSG_MAX_CELLTYPES];                /* enum lsg_ct_filter                  */
    uint32_t alt_bc[LSG_MAX_CELLTYPES][LSG_CALL_MAX_ALT];
    uint32_t alt_cc[LSG_MAX_CELLTYPES][LSG_CALL_MAX_ALT];
    int32_t  p_bc[LSG_MAX_CELLTYPES][LSG_CALL_MAX_ALT];   /* round(sf,4) * 1e4                   */
    int32_t  p_cc[LSG_MAX_CELLTYPES][LSG_CALL_MAX_ALT];
    uint32_t site_filter;               /* bit set, enum lsg_site_filter                        */
    int32_t  cell_types_min;            /* Cell_types_min_BC == Cell_types_min_CC               */
    int32_t  sum_alts_bc, sum_dp, sum_alts_cc, sum_nc;    /* Rest_BC / Rest_CC; sum_nc can be negative (step1.py:256) */
    int32_t  noise_p_bc, noise_p_cc;    /* round(1-cdf,4)*1e4; -1 when Sum_alts_bc == 0 (prints "1"); -2 = nan (negative n) */
    uint8_t  up_ctx[5], down_ctx[5];    /* ASCII; up_ctx[0]==0 => "." (POS < 6)                 */
    uint8_t  pad[2];
} lsg_call;

enum lsg_ct_filter { LSG_CF_NONE = 0, LSG_CF_NONSIG, LSG_CF_LOWSIG, LSG_CF_MULTI, LSG_CF_LOW_CELLS,
                     LSG_CF_LOW_READS, LSG_CF_PASS };
enum lsg_site_filter { LSG_SF_MULTIPLE_CELL_TYPES = 1, LSG_SF_MULTI_ALLELIC = 2, LSG_SF_MIN_CELL_TYPES = 4,
                       LSG_SF_CELL_TYPE_NOISE = 8, LSG_SF_NOISY_SITE = 16, LSG_SF_LC_UP = 32,
                       LSG_SF_LC_DOWN = 64, LSG_SF_CANDIDATE = 1u << 31 };

/* ---- lifetime -------------------------------------------------------------------------------*/
int         lsg_create(int device_id, lsg_ctx** out);
void        lsg_destroy(lsg_ctx* ctx);
const char* lsg_last_error(void);
const char* lsg_version(void);
/* Launch everything on this hipStream_t (pass torch.cuda.current_stream().cuda_stream);
 * NULL = the handle's own stream. */
int         lsg_set_stream(lsg_ctx* ctx, void* hip_stream);
int         lsg_synchronize(lsg_ctx* ctx);

/* ---- inputs ---------------------------------------------------------------------------------*/
/* Contig table = pysam.FastaFile.references / get_reference_length (BaseCellCounter.py:84-86). */
int lsg_set_contigs(lsg_ctx* ctx, int32_t n_contigs, const int64_t* lengths);
/* Upper-cased reference bases of one contig (inFasta.fetch(...).upper(), BaseCellCounter.py:202-203;
 * BaseCellCalling.step1.py:98-99). */
int lsg_load_reference(lsg_ctx* ctx, int32_t tid, const uint8_t* bases, int64_t len, int32_t on_device);
/* celltype_of[cb] in [0, n_celltypes) or 255 = barcode not used.  Replaces meta_to_dict + the
 * per-read routing of SplitBamCellTypes.py:16-36,83-90,173: a re-annotation pass only swaps this
 * table, the reads stay resident. */
int lsg_set_barcodes(lsg_ctx* ctx, const uint8_t* celltype_of, int32_t n_cb, int32_t n_celltypes);
/* Loads the read-record arrays: replaces reading the per-cell-type BAMs in run_interval (BaseCellCounter.py:190-191) AND the
 * pileup engine's transposition of reads into columns (bam.pileup -> htslib bam_plp, :191-198), done here once per load on the device.
 * The call builds the TILE STORE, the library's only resident form of the events: every (segment x 64-position tile) overlap of a
 * read that carries a barcode is one entry, the entries of a tile adjacent and sorted by barcode, eight entries to a 1 KB block held
 * transposed ([position][entry]); beside every entry its barcode, strand, SAM flag, MAPQ and owning read, so that every later
 * lsg_pileup_count (any parameters, any barcode table, any region) and lsg_genotype_cells run from the store alone.  Needs the
 * contigs (lsg_set_contigs).  The caller's arrays — host (on_device = 0) or device (1) — are free again when the call returns: the
 * per-read and per-segment arrays are copied, the events are read once.  Whatever offsets seg_ev_off holds is fine (segments may even
 * share events); a segment whose event range lies outside [0, n_events) or whose read index lies outside the reads is an error, and a
 * refused load leaves no reads behind.  Segments that do not lie inside their contig are never counted. */
int lsg_load_reads(lsg_ctx* ctx, const lsg_reads* reads);
/* Load filter: reads with MAPQ < min_mq, any SAM flag bit of flag_exclude, or (ignore_orphans) paired without proper pair are not
 * stored by the NEXT lsg_load_reads calls — what SplitBamCellTypes.py:110-113 does to the BAM before BaseCellCounter sees it
 * (LongSom passes the same min_MQ to both, R:SNVCalling.smk:22-27,52-59).  Default: none (0, 0, 0), every read with a barcode is
 * stored and any count parameters can be resolved later; with a filter the store is smaller by the dropped reads' share and a count or
 * genotyping pass whose own filters would admit a dropped read is refused.  The per-read arrays (depth cap, statistics) keep every read. */
int lsg_set_load_filter(lsg_ctx* ctx, int32_t min_mq, uint32_t flag_exclude, int32_t ignore_orphans);
/* What the caller knows about where the events of the NEXT lsg_load_reads calls lie: LSG_LAYOUT_PHASED = phased modulo 128 (lsg_reads above),
 * LSG_LAYOUT_COMPACT (default) = no promise.  A promise lets a load that keeps no store (lsg_set_store_policy) bin its entries by
 * 128-position windows; it is checked, and a load that finds it broken falls back by itself.  No reference counterpart. */
int lsg_set_events_layout(lsg_ctx* ctx, int32_t layout);
/* Device-side ingest of a whole BAM: the file's bytes in (host memory), the tile store out — BGZF inflate, record
 * chain, CB lookup, SplitBam's counters and the CIGAR walk all run on the GPU (csrc/ingest.hip), then lsg_load_reads on the device
 * arrays.  Replaces pysam.AlignmentFile + infile.fetch() + read.opt("CB") + the MAPQ counters of split_bam
 * (SplitBamCellTypes.py:51-124) and the record decode behind bam.pileup (BaseCellCounter.py:190-191).  Needs the contigs = the BAM
 * header's reference table (the caller parses the header: hostio.bam_header) and first_record_offset = the header's length in the
 * uncompressed stream.  barcodes: n_barcodes cleaned strings joined by '\n', ids[i] = dense id of barcode i (NULL: i); duplicated
 * strings: the last wins.  cb_pass / cb_low (may be NULL): per dense id, matched reads with MAPQ >= / < min_mapq (what the report of a
 * re-annotated table needs).  legacy_del_merge: htslib <= 1.10's 1D2D rule.  Returns -4 when the file's records
 * straddle its BGZF blocks so badly that the record chain did not settle (not written by htslib): decode such a file on the host. */
typedef struct {
    int64_t total_reads, pass_reads, cb_not_found, cb_not_matched, mapq_filtered;      /* {id}.report.txt, SplitBamCellTypes.py:181-187 */
    int64_t n_records, n_blocks, n_ubytes;
    float   ms_h2d, ms_inflate, ms_chain, ms_decode, ms_store, ms_total;               /* HIP-event / wall times of the phases */
    int32_t chain_rounds, pad_;
    int64_t last_key;           /* (tid << 32 | pos) of the last complete record, 2^63 - 1 for one without a reference, -1: no record */
} lsg_bam_info;
int lsg_load_bam(lsg_ctx* ctx, const uint8_t* file_bytes, int64_t n_bytes, int64_t first_record_offset, const char* barcodes, int32_t n_barcodes,
                 const int32_t* ids, int32_t min_mapq, int32_t legacy_del_merge, lsg_bam_info* info, int64_t* cb_pass, int64_t* cb_low, int64_t n_tally);
/* The same for a SLICE of a coordinate-sorted BAM — whole BGZF blocks cut out of the file, starting with a block in which a record starts
 * at first_record_offset (what the .bai's linear index gives: virtual offset = block start << 16 | offset in the block): what one rank of
 * a sharded run ingests, as the reference's workers fetch their window through the index (BaseCellCounter.py:190-191; the .bai is a
 * rule input, rules/SNVCalling.smk:6-7).  A record cut by the end of the slice is skipped.  SplitBam's counters and the per-barcode
 * tallies take only the records whose (tid << 32 | pos) lies in [count_lo_key, count_hi_key): neighbouring slices overlap (a rank
 * also needs the reads that reach into its region), the sum over the ranks counts every record once.  info->last_key tells the caller
 * whether the slice reached the first read starting at or after its region's end (every read before it in the file starts earlier:
 * nothing that overlaps the region is missing); if not it asks again with a longer slice (longsom_amd/regions.py). */
int lsg_load_bam_range(lsg_ctx* ctx, const uint8_t* slice_bytes, int64_t n_bytes, int64_t first_record_offset, const char* barcodes, int32_t n_barcodes,
                       const int32_t* ids, int32_t min_mapq, int32_t legacy_del_merge, int64_t count_lo_key, int64_t count_hi_key, lsg_bam_info* info,
                       int64_t* cb_pass, int64_t* cb_low, int64_t n_tally);
/* keep != 0: the next loads also keep a copy of the compact events (and seg_ev_off) beside the store, which is what
 * lsg_copy_reads_to_host returns (tests, sampling for a CPU baseline).  Default 0: the store is the only copy (2 B per event saved). */
int lsg_set_keep_reads(lsg_ctx* ctx, int32_t keep);
/* Gives back everything a load and its counts hold on the device (reads, the tile store, the build's cached temporaries, count rows,
 * call records): the handle is as after lsg_set_barcodes, ready for the next lsg_load_reads / lsg_load_bam.  What a worker process of the
 * reference does by ending (BaseCellCounter.py:392-402 starts a pool per BAM); here a long-lived handle moves from one sample to the next,
 * and a sample of C4's size (155 GB resident) needs the room of the one before it. */
int lsg_unload_reads(lsg_ctx* ctx);
/* The reference counts through bam.pileup(..., max_depth = 200000) (BaseCellCounter.py:191,
 * HCCVSingleCellGenotype.py:122): htslib stops admitting reads at a position while more than max_depth are
 * live in its buffer.  lsg_pileup_count models that cap exactly (lsg_count_params.max_depth); this call evaluates
 * the upper bound it uses to skip the work: the live reads of every cell-type BAM under the current barcode table
 * (reads of one cell type whose span touches a 64-position tile, maximum over tiles and cell types; before
 * lsg_set_barcodes: all reads with a barcode).  While the bound stays <= max_depth the cap cannot fire.
 * lsg_genotype_cells_grouped models it for the genotyping pileup of the unsplit BAM (HCCVSingleCellGenotype.py:122) with the all-reads
 * bound (lsg_max_live_reads_all).  Computed on request and cached until the reads or the barcode table change.  Returns the bound, -1 on
 * error. */
int64_t lsg_max_live_reads(lsg_ctx* ctx);
/* The same bound over EVERY resident read that carries a barcode, whatever its cell type (>= lsg_max_live_reads): what the
 * genotyping pileup of the unsplit BAM can hold at once (HCCVSingleCellGenotype.py:122) — a lower bound on it, since reads without a
 * usable CB tag are dropped at decode and not resident. */
int64_t lsg_max_live_reads_all(lsg_ctx* ctx);
/* The same question per cell type at POSITION resolution: the largest number of reads a pileup buffer can hold when a read is pushed,
 * counting that read (htslib refuses the push iff buffered + 1 > max_depth: a count with max_depth >= this value drops nothing).  The
 * tile-level bounds above over-count where many reads start and end inside one 64-position tile; lsg_pileup_count and the loads ask for
 * this one only when they cannot rule the cap out (a scan over the genome's positions per cell type: ~10 ms at C4).  Cached until the
 * reads or the barcode table change; -1 on error. */
int64_t lsg_max_live_reads_exact(lsg_ctx* ctx);
/* Restrict counting to the genomic region [ (tid_lo,pos_lo), (tid_hi,pos_hi) ) in (tid,pos) order;
 * positions must be multiples of 64.  This is how windows are sharded over GPUs: every rank loads
 * the reads overlapping its region (reads crossing a boundary are loaded by both ranks) and each
 * pileup column is counted by exactly one rank — the analogue of the reference's per-window
 * POS >= START and POS < END test (BaseCellCounter.py:200).  tid_hi == n_contigs, pos_hi == 0
 * means "to the end".  Reset with (0,0,n_contigs,0). */
int lsg_set_region(lsg_ctx* ctx, int32_t tid_lo, int64_t pos_lo, int32_t tid_hi, int64_t pos_hi);
/* The reference's pileup windows (BaseCellCounter.py --bin, default 50000; MakeWindows :81-113 cuts [1, 1 + bin), [1 + bin, ...) per contig
 * and run_interval opens a fresh bam.pileup per window, :185-191).  Result-neutral for the counts except through max_depth: every window's
 * pileup has a buffer of its own, so lsg_pileup_count replays the cap per window, and the loads that follow never let a resident entry
 * cross a window edge (a read dropped in one window may be counted in the next).  window >= 64; default 50000. */
int lsg_set_pileup_window(lsg_ctx* ctx, int32_t window);
/* One pass for a BAM's load AND its first count.  A rule of the reference counts every BAM exactly once
 * (bam.pileup over all windows, BaseCellCounter.py:182-320, after SplitBamCellTypes.py:65-124 routed the reads); when the
 * count's parameters are known while the reads are loaded - they are in every fused rule - the loads that follow
 * (lsg_load_reads, lsg_load_bam, lsg_load_bam_range) count in the same pass that lays the events out per column: the wave that
 * transposes a block of the store adds its events into the per-position counters while it holds them.  Requires the barcode
 * table (lsg_set_barcodes, at most two cell types), every reference and the region to be set before the load; a load for which
 * the pileup's depth cap could fire (lsg_max_live_reads_all() >= max_depth) is made without the count.  The first
 * lsg_pileup_count after the load with EQUAL parameters, the same barcode table and region returns that count's result without
 * another pass; any other call counts over the resident store as always.  params == NULL switches it off (default). */
int lsg_set_count_at_load(lsg_ctx* ctx, const lsg_count_params* params);
/* What the loads that follow keep beside the count they make (lsg_set_count_at_load).  LSG_STORE_KEEP (default): the tile store, the
 * resident form every later count, region, barcode table and genotyping pass works on.  LSG_STORE_SKIP_WHEN_COUNTED: a rule of the
 * reference reads a BAM, counts it once and is done (BaseCellCounter.py:182-320: one bam.pileup per window, nothing is kept) - when the
 * load can make its count in its own pass (the conditions of lsg_set_count_at_load) it counts straight from the caller's events and
 * writes no store: what derives from that count (lsg_fetch_counts, the exports, lsg_call_step1 ...) works as always, lsg_pileup_count
 * with the load's parameters returns it, and everything that needs the store (another count, lsg_genotype_cells) fails with a message
 * until reads are loaded again.  A load that cannot make its count (the depth cap could fire, more than two cell types) builds the
 * store as under LSG_STORE_KEEP.  When the count's read filters are the load's (lsg_set_load_filter with the count's min_mq,
 * flag_exclude and ignore_orphans: every stored read is admitted) and the tiles rule the depth cap out, such a load carries 8-byte keys
 * alone through its scatter and sort instead of 12-byte key + value pairs (store.hip keys_only): the same count, a tenth faster. */
/* The BAM loads that follow (lsg_load_bam, lsg_load_bam_range) keep the reads without a CB tag or with a barcode that is not listed
 * (cb = -1) instead of dropping them at decode time.  Such a read is never counted (SplitBamCellTypes.py:74-90 routes it nowhere), but
 * the per-cell genotyping piles up the UNSPLIT BAM (HCCVSingleCellGenotype.py:121-122): every read that passes that pileup's own
 * filters takes a place in its max_depth buffer, listed or not, and lsg_genotype_cells_grouped replays the buffer over the resident
 * reads.  Default off (the counting rules never see these reads; their events cost memory).  The host decoder's twin is
 * lsio_set_keep_unlisted. */
int lsg_set_keep_unlisted(lsg_ctx* ctx, int32_t on);
enum { LSG_STORE_KEEP = 0, LSG_STORE_SKIP_WHEN_COUNTED = 1 };
int lsg_set_store_policy(lsg_ctx* ctx, int32_t policy);

/* ---- hot path -------------------------------------------------------------------------------*/
/* Per-cell-type pileup base counting over every covered column of the loaded reads; replaces
 * split_bam's filter (SplitBamCellTypes.py:65-124), run_interval (BaseCellCounter.py:182-320) for
 * all windows, and the gates at :211,:221,:282,:294.  Results stay on the device.
 * n_rows[c] = emitted sites of cell type c; *n_columns = pileup columns with >=1 counted entry,
 * summed over cell types (the "genomic sites" of BASELINE.json's metric). */
int lsg_pileup_count(lsg_ctx* ctx, const lsg_count_params* params, int64_t* n_rows, int64_t* n_columns);
/* Copies cell type ct's rows to the host in genomic order (tid, pos ascending):
 * keys[n] = (tid<<32)|pos0, ref[n] = reference base, counts[n*42].  The resident rows keep 34 of the 42
 * words; BCr[8] is delivered as BC - BCf (what BaseCellCounter.py:271-279 counts: every read is forward or reverse). */
int lsg_fetch_counts(lsg_ctx* ctx, int32_t ct, int64_t* keys, uint8_t* ref, uint32_t* counts, int64_t capacity);

/* Installs per-cell-type count rows (keys[c][i] = (tid<<32)|pos0 strictly ascending, counts[c] =
 * n_rows[c] x 42 words) as if lsg_pileup_count had produced them: the file-level entry of the
 * MergeCounts / BaseCellCalling_step1 rules, whose inputs are BaseCellCounter TSVs
 * (R:SNVCalling.smk:62-156; merge_cell_types_files reads them at MergeBaseCellCounts.py:139-161). */
int lsg_load_counts(lsg_ctx* ctx, int32_t n_celltypes, const int64_t* const* keys, const uint32_t* const* counts,
                    const int64_t* n_rows);

/* Outer join of the per-cell-type rows on (tid,pos) + step-1 arithmetic on every merged site;
 * replaces merge_cell_types_files (MergeBaseCellCounts.py:116-204) and variant_calling_step1
 * (BaseCellCalling.step1.py:19-476).  *n_sites = merged sites, *n_candidates = sites with ALT != ".". */
int lsg_call_step1(lsg_ctx* ctx, const lsg_call_params* params, int64_t* n_sites, int64_t* n_candidates);
/* Copies call records in genomic order; candidates_only != 0 keeps rows with ALT != "." or FILTER != "." */
int lsg_fetch_calls(lsg_ctx* ctx, lsg_call* out, int64_t capacity, int32_t candidates_only, int64_t* n_out);

/* Compacts call records, in genomic order, into a caller-owned DEVICE buffer (e.g. a torch tensor
 * handed to an RCCL all-gather): kind 0 = every site, 1 = rows step 2 keeps (ALT != "." or
 * FILTER != "."), 2 = PASS candidates only (no site filter and a PASS cell type).  dst_device may be
 * NULL to only count; *n_out receives the number of selected rows. */
int lsg_export_calls(lsg_ctx* ctx, int32_t kind, void* dst_device, int64_t capacity, int64_t* n_out);

/* ---- the tables' text, printed on the device -------------------------------------------------------
 * The rows of the reference's three big tables, byte for byte what its writers print:
 *   LSG_TABLE_COUNTS + c  <sample>.<cell type c>.tsv rows       BaseCellCounter.py:300-308 (one row per count row of cell type c)
 *   LSG_TABLE_MERGED      BaseCellCounts.AllCellTypes.tsv rows  MergeBaseCellCounts.py:59-84,116-204 (outer join, "NA" for an absent cell type)
 *   LSG_TABLE_STEP1       calling.step1.tsv rows                BaseCellCalling.step1.py:430-476 (one row per merged site)
 *   LSG_TABLE_STEP1_KEPT  the rows of that table step 2's awk filter keeps (ALT != "." and FILTER != ".", BaseCellCalling.step2.py:23)
 *   LSG_TABLE_STEP2       calling.step2.tsv rows when step 2 has no gnomAD source and --min_distance 0 (LongSom's setting): the kept rows,
 *                         FILTER tagged "RNA_editing_db" / "PoN_SR" / "PoN_LR" by the resident position sets (lsg_load_posset; GetExtraFilters,
 *                         BaseCellCalling.step2.py:142-158; a tag replaces a bare "PASS"), "NA" cells empty (step2.py's pandas round trip)
 *   LSG_TABLE_STEP3_ROWS  the rows of LSG_TABLE_STEP2 step 3 can keep (made by lsg_step2_summary, not by lsg_format_table)
 * Rows only (the header lines are the caller's), in the reference's order: contigs in Python string order, positions ascending.
 * At 24 M sites these are 17 GB of text: a kernel prints them from the count rows and call records where they lie (two passes:
 * lengths, then bytes), the host only moves bytes.  The merged and step-1 tables need lsg_call_step1 (its merged site list). */
enum lsg_table { LSG_TABLE_COUNTS = 0, LSG_TABLE_MERGED = LSG_MAX_CELLTYPES, LSG_TABLE_STEP1, LSG_TABLE_STEP1_KEPT, LSG_TABLE_STEP2, LSG_TABLE_STEP3_ROWS, LSG_TABLE_SLOTS };
/* Names the rows print: contig_names / celltype_names are '\n'-joined, in lsg_set_contigs / cell-type index order. */
int lsg_set_table_names(lsg_ctx* ctx, int32_t n_contigs, const char* contig_names, int32_t n_celltypes, const char* celltype_names);
/* Prints one table into a device buffer the handle keeps for it (until lsg_free_table or the next format of the same table);
 * *n_bytes = size of the text.  The text is a snapshot: later counts and calls do not change it. */
int lsg_format_table(lsg_ctx* ctx, int32_t table, int64_t* n_bytes);
/* The formatted text into host memory (capacity >= its size). */
int lsg_copy_table(lsg_ctx* ctx, int32_t table, char* dst_host, int64_t capacity);
/* Appends the formatted text to the file at `path` (created if absent): device -> pinned staging -> pwrite, pipelined on a stream of
 * its own.  Only reads the table's buffer: may run on another host thread beside any other call on the handle except
 * lsg_format_table / lsg_free_table of the same table and lsg_destroy. */
int lsg_append_table(lsg_ctx* ctx, int32_t table, const char* path);
/* What BaseCellCalling.step3.py needs of the whole step-2 table before it parses anything, read off LSG_TABLE_STEP2's text on the device
 * (directly after its lsg_format_table): kinds[c] = which kinds of cell column c of n_cols holds over ALL rows, as the bits 1 missing
 * value, 2 integer, 4 float, 8 a number pandas would print differently, 16 anything else (pandas infers a column's dtype over the whole
 * file: step3.py:41 reads it all; bits other than 16 mean nothing in a column that has 16); and, as table LSG_TABLE_STEP3_ROWS of
 * *n_survivor_bytes bytes, the rows whose FILTER holds none of the patterns step 3 drops rows by (step3.py:49-52 for chrM rows, :60-84
 * for the others) and whose Cell_types is not "Non-Cancer" - the only rows step 3 has to parse. */
int lsg_step2_summary(lsg_ctx* ctx, int32_t n_cols, uint8_t* kinds, int64_t* n_survivor_bytes);
/* Frees one table's text (table < 0: every table's, and the flat row copies they were printed from). */
int lsg_free_table(lsg_ctx* ctx, int32_t table);

/* Position sets (RNA-editing / PoN_SR / PoN_LR; build_dict, BaseCellCalling.step2.py:197-221):
 * sorted unique keys (tid<<32)|pos1 resident in HBM; kind in [0,3). */
int lsg_load_posset(lsg_ctx* ctx, int32_t kind, const int64_t* keys, int64_t n, int32_t on_device);
/* Membership of n query keys in set `kind` (GetExtraFilters, step2.py:142-158). hits[i] = 0/1. */
int lsg_probe_posset(lsg_ctx* ctx, int32_t kind, const int64_t* keys, int64_t n, uint8_t* hits, int32_t on_device);

/* ---- per-cell genotyping at target sites ---------------------------------
 * HCCVSingleCellGenotype.py:82-220 (twin: SNVCalling/SingleCellGenotype.py:84-228): for every target
 * site and every barcode of barcodes.tsv
 *   Dp  = pileup entries of that cell at the site whose symbol is one of A,C,T,G,I,D,N (:147-149; 'O'
 *         and 'NA' are not counted) and whose base quality is >= min_bq (pileup min_base_quality, :123),
 *   Alt = those whose symbol equals the site's expected alt (:171-175).
 * alt_only != 0 is --alt_flag Alt (:150-151): only reads carrying the expected alt are looked at.
 * Read admission: pileup's flag filter and ignore_orphans, min_mq (:123), not secondary / duplicate /
 * supplementary (:168), CB present and listed (:160-164); strict_cb != 0 also drops reads whose raw CB
 * tag carried a "-suffix" (the reference looks the raw tag up in the cleaned table, :160-161;
 * liblongsom_io marks such reads with LSG_FLAG_CB_SUFFIX in read_flag).
 * site_keys = (tid << 32) | pos0, strictly ascending; alt_sym = symbol class 0..6 per site (255 = none).
 * dp / alt: [n_sites][n_cb] uint32, zeroed by the call.  Needs contigs, barcodes and reads. */
#define LSG_FLAG_CB_SUFFIX 0x8000u
typedef struct {
    int32_t  min_bq, min_mq;
    uint32_t flag_exclude;      /* 0xF04 */
    int32_t  ignore_orphans;
    int32_t  alt_only;
    int32_t  strict_cb;
} lsg_genotype_params;
int lsg_genotype_cells(lsg_ctx* ctx, const lsg_genotype_params* params, int64_t n_sites, const int64_t* site_keys,
                       const uint8_t* alt_sym, uint32_t* dp, uint32_t* alt, int32_t on_device);
/* The same with the pileup's depth cap (HCCVSingleCellGenotype.py:122: bam.pileup(CHROM, START, END, ..., max_depth = 200000), one call
 * per window of target sites, build_dict_variants :245-265): the sites [group_off[g], group_off[g + 1]) are one such window (same
 * contig), its region [first site - 1, last site + 1).  For every window whose region can hold more than max_depth reads, htslib's rule
 * (a read that is not the first of its start position is dropped while the buffer exceeds max_depth; lsg_count_params.max_depth states
 * it) is replayed over the resident reads that overlap the region and the reads it drops are left out of that window's sites.  The
 * stream is the reads of the resident load — all cell types, reads with a listed barcode (the decoders keep no others): for a BAM
 * whose reads all carry listed barcodes this is the reference's buffer, otherwise a lower bound on it.  max_depth <= 0 or n_groups == 0:
 * lsg_genotype_cells.  Free while lsg_max_live_reads_all() stays below the cap. */
int lsg_genotype_cells_grouped(lsg_ctx* ctx, const lsg_genotype_params* params, int32_t max_depth, int64_t n_sites, const int64_t* site_keys,
                               const uint8_t* alt_sym, int64_t n_groups, const int64_t* group_off, uint32_t* dp, uint32_t* alt, int32_t on_device);
/* round(betabinom.sf(k - 0.001, n, alpha, beta), 4) * 1e4 for n_items integer (k, n) pairs, evaluated on the
 * device with the step-1 tail code (HCCVSingleCellGenotype.py:204; k[i] <= 0 gives 10000). */
int lsg_betabinom_sf4(lsg_ctx* ctx, int64_t n_items, const uint32_t* k, const uint32_t* n, double alpha, double beta, int32_t* out_p4);
/* The same tails with the UNROUNDED fp64 value beside the rounded one: what the exactness audit measures (how far every p of a
 * workload lies from a 4-decimal rounding tie of round(betabinom.sf(...), 4), BaseCellCalling.step1.py:196,201; tools/p_margins.py). */
int lsg_betabinom_sf(lsg_ctx* ctx, int64_t n_items, const uint32_t* k, const uint32_t* n, double alpha, double beta, int32_t* out_p4, double* out_p);

/* ---- measurement helpers --------------------------------------------------------------------*/
/* Statistics of the last lsg_pileup_count: admitted reads / segments / events (events that passed
 * read admission, before the base-quality gate), tile entries, non-empty units, deep units. */
typedef struct {
    int64_t n_reads_admitted, n_segs_admitted, n_events_admitted;
    int64_t n_entries, n_units, n_deep_units;
    int64_t n_events_wave, n_events_deep;  /* events loaded by k_pileup_wave / k_pileup_deep      */
    int64_t n_rows_wave, n_rows_deep;      /* rows emitted by each kernel (all cell types)         */
    float   ms_bin, ms_deep, ms_wave, ms_total;   /* HIP-event times of the last call             */
    float   ms_walk;                       /* k_walk_block alone (HIP events around the launch)    */
    float   pad_;
    int64_t rows_by_kernel[4];             /* rows emitted by: 0 k_pileup_wave, 1 k_walk_block, 2 k_pileup_huge, 3 k_finalize_multi */
    int64_t events_by_kernel[4];           /* events loaded by the same kernels (3 = 0)            */
} lsg_count_stats;
int lsg_get_count_stats(lsg_ctx* ctx, lsg_count_stats* out);

/* What the load's tile store cost.  path: 2 = the store was built by the load alone, 3 = in the pass that also made the first count
 * (lsg_set_count_at_load), 4 = the load made its count and kept no store (lsg_set_store_policy), 5 = the same over tile-phased events
 * (LSG_LAYOUT_PHASED: every entry fetched as its one 128-byte line).  build_ms: wall time
 * lsg_load_reads spent building the store (device kernels + their host synchronisations).  store_bytes: device memory the store, its
 * per-read / per-segment arrays, kept events and cached build temporaries hold.  No reference counterpart: the reference re-reads
 * the BAM per window (BaseCellCounter.py:198-225). */
int lsg_get_layout_info(lsg_ctx* ctx, int32_t* path, double* build_ms, int64_t* store_bytes);
/* HIP-event times (ms) of the last load's build phases: [0] capacities + scatter, [1] sort of every tile's entries by barcode,
 * [2] per-entry words, [3] event gather into the transposed blocks (the build's dominant kernel, k_tm_gather). */
int lsg_get_build_times(lsg_ctx* ctx, float* ms4);
/* Shape of the resident tile store: entries ((segment x tile) overlaps of reads with a barcode), 1 KB blocks, and the events the
 * entries hold (what k_tm_gather reads once from the caller's array and writes once: its algorithmic bytes are 4 per event). */
int lsg_get_store_shape(lsg_ctx* ctx, int64_t* n_entries, int64_t* n_blocks, int64_t* n_events);

/* (The synthetic workload of bench.py and the tests - lsg_synth_*, the read-back of resident arrays - is declared in
 * include/longsom_synth.h: measurement and test support, not part of the boundary a rule's script binds.) */

#ifdef __cplusplus
}
#endif
#endif /* LONGSOM_HIP_H */
