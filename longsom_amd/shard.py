"""Sharding of the hot path over the GPUs of one node.

Pileup columns are independent, so genomic REGIONS are sharded: every rank loads the reads overlapping
its region (reads crossing a boundary are loaded by both neighbours), counts only its own columns
(lsg_set_region) and runs merge + step 1 on them.  The only exchange is an all-gather of the ranks'
PASS-candidate call rows (what step 3's 10 kb cluster filter has to see across boundaries); full count
tables never travel — each rank writes its own slice of the TSVs.  torch.distributed with backend
"nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from typing import List, Tuple

import numpy as np

import os

from . import synth

CALL_BYTES = 336          # sizeof(lsg_call)
COLUMN_COST = float(os.environ.get("LSG_COLUMN_COST", "0.036"))       # one emitted (site, cell type) column costs about this many reads of kernel time (MI355X, C2)


def region_shards(model, world: int) -> List[Tuple[Tuple[int, int], Tuple[int, int], int, int]]:
    """Tile-aligned region boundaries at gene starts that balance the read count (event count: read lengths are
    iid), and for every rank the contiguous gene range containing all genes that overlap its region.
    Returns [(lo=(tid,pos), hi=(tid,pos), gene_lo, gene_hi)] with regions half-open in (tid,pos) order."""
    g0 = model.gene_exon_off[:-1]; g1 = model.gene_exon_off[1:] - 1
    tid = model.gene_tid.astype(np.int64)
    start = model.exon_start[g0].astype(np.int64)
    end = (model.exon_start[g1] + model.exon_len[g1]).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(model.contig_len)])[:-1]
    lin_s, lin_e = off[tid] + start, off[tid] + end
    # cost of a gene = its reads (events: read lengths are iid) + its pileup columns (~2 cell types x exon bases), the
    # latter weighted by the measured per-column / per-read kernel time ratio (row emission + call stage vs walk)
    exon_bases = np.add.reduceat(model.exon_len.astype(np.int64), model.gene_exon_off[:-1]) if model.n_genes else np.zeros(0, np.int64)
    cost = np.diff(model.gene_read_off).astype(np.float64) + COLUMN_COST * 2.0 * exon_bases
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = float(cum[-1])
    bounds = [(0, 0)]
    for r in range(1, world):
        g = int(np.searchsorted(cum, total * r / world, side="left"))
        g = min(max(g, 0), model.n_genes - 1)
        bounds.append(max((int(tid[g]), int(start[g]) // 64 * 64), bounds[-1]))
    bounds.append((len(model.contig_len), 0))
    genome = int(np.sum(model.contig_len))
    shards = []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        lin_lo = int(off[lo[0]]) + lo[1] if lo[0] < len(off) else genome
        lin_hi = int(off[hi[0]]) + hi[1] if hi[0] < len(off) else genome
        ov = np.nonzero((lin_e > lin_lo) & (lin_s < lin_hi))[0]
        g_lo, g_hi = (int(ov.min()), int(ov.max()) + 1) if len(ov) else (0, 0)
        shards.append((lo, hi, g_lo, g_hi))
    return shards


def sub_model(model, g_lo: int, g_hi: int):
    """The workload model restricted to genes [g_lo, g_hi); draws stay keyed by the global read index."""
    x0, x1 = int(model.gene_exon_off[g_lo]), int(model.gene_exon_off[g_hi])
    offr = model.gene_read_off[g_lo:g_hi + 1] - model.gene_read_off[g_lo]
    return synth.SynthModel(model.seed, model.contig_names, model.contig_len, model.gene_tid[g_lo:g_hi].copy(),
                            (model.gene_exon_off[g_lo:g_hi + 1] - x0).astype(np.int32), model.exon_start[x0:x1].copy(),
                            model.exon_len[x0:x1].copy(), model.exon_cum[x0:x1].copy(), offr.astype(np.int64), model.celltype_of,
                            int(offr[-1]) if len(offr) else 0, model.n_cb, model.snp_mod, int(model.read_base + model.gene_read_off[g_lo]), model.layout)


def in_region(keys: np.ndarray, lo, hi) -> np.ndarray:
    """mask of (tid<<32|pos) keys inside the half-open region [lo, hi)"""
    klo = (lo[0] << 32) | lo[1]
    khi = (hi[0] << 32) | hi[1]
    return (keys >= klo) & (keys < khi)


def all_gather_rows(local, dist, device=None):
    """All-gather of a variable number of fixed-size rows per rank (uint8 tensor [n, row_bytes]): counts first,
    then one all-gather on buffers padded to the largest count.  Returns the list of per-rank row tensors."""
    import torch
    world = dist.get_world_size()
    row = local.shape[1]
    dev = device if device is not None else local.device
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, cnt)
    mx = max(1, max(int(c.item()) for c in counts))
    pad = torch.zeros((mx, row), dtype=torch.uint8, device=dev)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros((mx, row), dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[: int(c.item())] for b, c in zip(bufs, counts)]
