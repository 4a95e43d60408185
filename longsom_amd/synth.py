"""Synthetic workload of BASELINE.md §4 / SURVEY.md §8(d): gene tables + expression -> lsg_synth_model.

The per-read / per-base draws live in longsom_amd/csrc/synth_model.h (one source compiled for the
GPU generator and for the host BAM writer); this module builds the deterministic tables they use:
contigs (hg38 lengths / scale), genes (2-12 exons of 80-400 bp, introns 100-20 000 bp), a Zipf(1.1)
expression profile with chrM at a fixed share and a per-gene depth cap (< 200 000, so the reference's
max_depth rule is never triggered), and the cancer / non-cancer barcode split.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

HG38_LEN = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
    "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
    "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
    "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415, "chrM": 16569,
}

# named configurations (BASELINE.json configs[0..3]); scale divides the nuclear contig lengths
CONFIGS = {
    "C1": dict(seed=22, contigs=["chr22"], scale=10, n_genes=800, n_reads=50_000, n_cb=200, chrm_share=0.0),
    "C2": dict(seed=2, contigs=list(HG38_LEN), scale=10, n_genes=20_000, n_reads=10_000_000, n_cb=5_000, chrm_share=0.08),
    "C4": dict(seed=4, contigs=list(HG38_LEN), scale=10, n_genes=20_000, n_reads=50_000_000, n_cb=20_000, chrm_share=0.08),
}


class SynthModelC(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_reads", C.c_int64), ("read_base", C.c_int64),
        ("n_genes", C.c_int32), ("n_cb", C.c_int32), ("n_contigs", C.c_int32), ("snp_mod", C.c_int32),
        ("gene_tid", C.c_void_p), ("gene_exon_off", C.c_void_p), ("exon_start", C.c_void_p), ("exon_len", C.c_void_p),
        ("exon_cum", C.c_void_p), ("gene_read_off", C.c_void_p), ("celltype_of", C.c_void_p),
        ("layout", C.c_int32), ("pad_", C.c_int32),
    ]


@dataclass
class SynthModel:
    seed: int
    contig_names: list
    contig_len: np.ndarray      # int64
    gene_tid: np.ndarray        # int32 [G]
    gene_exon_off: np.ndarray   # int32 [G+1]
    exon_start: np.ndarray      # int32
    exon_len: np.ndarray        # int32
    exon_cum: np.ndarray        # int32
    gene_read_off: np.ndarray   # int64 [G+1]
    celltype_of: np.ndarray     # uint8 [n_cb]  0 = Cancer, 1 = Non-Cancer
    n_reads: int
    n_cb: int
    snp_mod: int = 15000
    read_base: int = 0
    layout: int = 0             # LSG_LAYOUT_COMPACT; 1 = LSG_LAYOUT_PHASED (include/longsom_hip.h: every read-in-a-tile inside one 128-byte line)

    @property
    def n_genes(self): return len(self.gene_tid)

    def as_c(self) -> SynthModelC:
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        return SynthModelC(self.seed, self.n_reads, self.read_base, self.n_genes, self.n_cb, len(self.contig_len), self.snp_mod,
                           p(self.gene_tid), p(self.gene_exon_off), p(self.exon_start), p(self.exon_len), p(self.exon_cum),
                           p(self.gene_read_off), p(self.celltype_of), int(self.layout), 0)


def build_model(seed, contigs, scale, n_genes, n_reads, n_cb, chrm_share, cancer_frac=0.4, depth_cap=150_000,
                snp_mod=15000) -> SynthModel:
    rng = np.random.Generator(np.random.PCG64(seed))
    names = list(contigs)
    lens = np.array([HG38_LEN[c] if c == "chrM" else HG38_LEN[c] // scale for c in names], dtype=np.int64)
    nuclear = [i for i, c in enumerate(names) if c != "chrM"]
    has_m = "chrM" in names
    n_m = 13 if has_m else 0
    n_nuc = n_genes - n_m
    # nuclear genes: contig ~ length, 2-12 exons
    w = lens[nuclear].astype(np.float64); w /= w.sum()
    g_tid = rng.choice(np.array(nuclear), size=n_nuc, p=w)
    n_ex = rng.integers(2, 13, size=n_nuc)
    genes = []
    for g in range(n_nuc):
        k = int(n_ex[g])
        ex_len = np.clip(np.exp(rng.normal(np.log(160.0), 0.5, size=k)), 80, 400).astype(np.int64)
        intr = rng.integers(100, 20001, size=k - 1)
        span = int(ex_len.sum() + intr.sum())
        clen = int(lens[g_tid[g]])
        if span >= clen - 2000:       # tiny contigs: shrink introns
            intr = np.full(k - 1, 100); span = int(ex_len.sum() + intr.sum())
        start = int(rng.integers(1000, max(1001, clen - span - 1000)))
        starts = start + np.concatenate([[0], np.cumsum(ex_len[:-1] + intr)])
        genes.append((int(g_tid[g]), starts.astype(np.int64), ex_len))
    if has_m:
        tid_m = names.index("chrM")
        # 13 non-overlapping single-exon genes tiling chrM
        edges = np.linspace(50, HG38_LEN["chrM"] - 50, n_m + 1).astype(np.int64)
        for k in range(n_m):
            genes.append((tid_m, np.array([edges[k]]), np.array([edges[k + 1] - edges[k] - 10])))
    # expression: Zipf(1.1) over a random ranking of nuclear genes; chrM genes share chrm_share equally
    ranks = rng.permutation(n_nuc) + 1
    expr = np.concatenate([ranks.astype(np.float64) ** -1.1, np.zeros(n_m)])
    expr[:n_nuc] *= (1.0 - (chrm_share if has_m else 0.0)) / expr[:n_nuc].sum()
    if has_m:
        expr[n_nuc:] = chrm_share / n_m
    # depth cap: a gene's reads can all cover the same column
    cap = depth_cap / max(1, n_reads)
    for _ in range(50):
        over = expr > cap
        if not over.any():
            break
        excess = (expr[over] - cap).sum()
        expr[over] = cap
        free = ~over
        expr[free] += excess * expr[free] / expr[free].sum()
    # sort genes by (tid, start): read index order ~ coordinate order
    order = sorted(range(len(genes)), key=lambda g: (genes[g][0], int(genes[g][1][0])))
    genes = [genes[g] for g in order]
    expr = expr[order]
    cum = np.floor(np.cumsum(expr) * n_reads + 0.5).astype(np.int64)
    cum[-1] = n_reads
    gene_read_off = np.concatenate([[0], cum]).astype(np.int64)
    gene_tid = np.array([g[0] for g in genes], dtype=np.int32)
    gene_exon_off = np.concatenate([[0], np.cumsum([len(g[1]) for g in genes])]).astype(np.int32)
    exon_start = np.concatenate([g[1] for g in genes]).astype(np.int32)
    exon_len = np.concatenate([g[2] for g in genes]).astype(np.int32)
    exon_cum = np.concatenate([np.concatenate([[0], np.cumsum(g[2])[:-1]]) for g in genes]).astype(np.int32)
    celltype_of = np.ones(n_cb, dtype=np.uint8)
    celltype_of[: int(round(n_cb * cancer_frac))] = 0
    return SynthModel(seed, names, lens, gene_tid, gene_exon_off, exon_start, exon_len, exon_cum, gene_read_off,
                      celltype_of, int(n_reads), int(n_cb), int(snp_mod))


def named(config: str, layout: int = 0, **override) -> SynthModel:
    kw = dict(CONFIGS[config]); kw.update(override)
    m = build_model(**kw)
    m.layout = int(layout)
    return m


def shard_reads(model: SynthModel, rank: int, world: int) -> SynthModel:
    """Contiguous gene range with ~1/world of the reads (event-balanced: read length is iid).
    Draws stay keyed by the global read index (read_base), so the union of the shards' reads is
    exactly the unsharded model's read set."""
    tot = model.n_reads
    lo_t, hi_t = tot * rank // world, tot * (rank + 1) // world
    g_lo = int(np.searchsorted(model.gene_read_off, lo_t, side="right") - 1) if rank > 0 else 0
    g_hi = int(np.searchsorted(model.gene_read_off, hi_t, side="right") - 1) if rank < world - 1 else model.n_genes
    g_lo = max(0, min(g_lo, model.n_genes)); g_hi = max(g_lo, min(g_hi, model.n_genes))
    x0, x1 = int(model.gene_exon_off[g_lo]), int(model.gene_exon_off[g_hi])
    off = model.gene_read_off[g_lo:g_hi + 1] - model.gene_read_off[g_lo]
    return SynthModel(model.seed, model.contig_names, model.contig_len, model.gene_tid[g_lo:g_hi].copy(),
                      (model.gene_exon_off[g_lo:g_hi + 1] - x0).astype(np.int32), model.exon_start[x0:x1].copy(),
                      model.exon_len[x0:x1].copy(), model.exon_cum[x0:x1].copy(), off.astype(np.int64),
                      model.celltype_of, int(off[-1]) if len(off) else 0, model.n_cb, model.snp_mod,
                      int(model.read_base + model.gene_read_off[g_lo]), model.layout)
