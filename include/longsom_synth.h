/* longsom_synth.h - measurement and test support of liblongsom_hip.so: the synthetic long-read workload of BASELINE.md section 4 generated
 * straight into HBM (bench.py's inputs never cross PCIe), and the read-back of a handle's resident arrays for the CPU oracle.  NOT part of
 * the drop-in boundary (include/longsom_hip.h): nothing a rule's script binds is declared here, and the reference has no counterpart. */
#ifndef LONGSOM_SYNTH_H
#define LONGSOM_SYNTH_H
#include "longsom_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic workload (bench / tests; not part of the reference's path) --------------------*/
/* Gene/expression tables of the BASELINE.md §4 workload model (built by longsom_amd/synth.py; the
 * per-read / per-base draws are counter-based hashes, see longsom_amd/csrc/synth_model.h).
 * All pointers are HOST pointers; celltype_of is ignored on the device path (lsg_set_barcodes's
 * table is used). */
typedef struct {
    uint64_t seed;
    int64_t  n_reads;        /* reads of this (shard of the) model                                 */
    int64_t  read_base;      /* global index of local read 0: draws are keyed by read_base + i, so a
                                shard generates exactly the reads the unsharded model would          */
    int32_t  n_genes, n_cb, n_contigs, snp_mod;
    const int32_t* gene_tid;        /* [G]                                   */
    const int32_t* gene_exon_off;   /* [G+1] index into the exon arrays      */
    const int32_t* exon_start;      /* 0-based reference start of each exon  */
    const int32_t* exon_len;
    const int32_t* exon_cum;        /* transcript coordinate of the exon's first base */
    const int64_t* gene_read_off;   /* [G+1] reads [off[g], off[g+1]) belong to gene g */
    const uint8_t* celltype_of;     /* [n_cb] 0 = Cancer                     */
    int32_t  layout;                /* where the generated events lie: LSG_LAYOUT_COMPACT or LSG_LAYOUT_PHASED (lsg_reads above) */
    int32_t  pad_;
} lsg_synth_model;

/* Fills the reference of every contig with the model's synthetic genome, in HBM. */
int lsg_synth_reference(lsg_ctx* ctx, uint64_t seed);
/* Generates the model's read-record arrays directly in HBM and makes them the loaded reads (generate + lsg_load_reads). */
int lsg_synth_reads(lsg_ctx* ctx, const lsg_synth_model* model);
/* Only generates them: *out describes compact device arrays owned by the handle (valid until the next generate, lsg_synth_reads or
 * lsg_destroy) — a stand-in for a caller whose decoded BAM is device-resident; bench.py times lsg_load_reads on them. */
int lsg_synth_generate(lsg_ctx* ctx, const lsg_synth_model* model, lsg_reads* out);
int lsg_get_reads_shape(lsg_ctx* ctx, int64_t* n_reads, int64_t* n_segs, int64_t* n_events);
/* Copies the resident read-record arrays / reference into caller-allocated host arrays.  out->events and out->seg_ev_off both NULL:
 * the per-read and per-segment arrays only (always resident: what a sharded run cuts its regions from); otherwise the events too, which
 * are there only when the load kept them (lsg_set_keep_reads). */
int lsg_copy_reads_to_host(lsg_ctx* ctx, const lsg_reads* out);
int lsg_copy_reference_to_host(lsg_ctx* ctx, int32_t tid, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* LONGSOM_SYNTH_H */
