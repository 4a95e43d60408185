"""Development aid / DESIGN.md numbers: lsg_genotype_cells (per-cell genotyping at target sites) on the C2
workload, n target sites drawn from the exons; the matrices come back to the host ([n_sites][n_cb] uint32 x 2)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
model = synth.named("C2", n_reads=n_reads)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
rng = np.random.default_rng(3)
x = rng.choice(len(model.exon_start), size=n_sites, replace=True)
gene_of_exon = np.searchsorted(model.gene_exon_off, x, side="right") - 1
keys = np.unique((model.gene_tid[gene_of_exon].astype(np.int64) << 32) | (model.exon_start[x].astype(np.int64) + rng.integers(0, np.maximum(model.exon_len[x], 1))))
alt = rng.integers(0, 4, len(keys)).astype(np.uint8)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); dp, al = eng.genotype_cells(keys, alt); best = min(best, time.perf_counter() - t0)
n_reads_, n_segs, n_ev = eng.reads_shape()
stream = n_segs * 20 + n_reads_ * 11
print("sites %d cells %d: %.2f ms per call (incl. the %.0f MB of matrices to the host); segment stream %.2f GB -> %.0f GB/s; covered (site, cell) pairs %d, alt pairs %d, reads counted %d"
      % (len(keys), dp.shape[1], best * 1e3, dp.nbytes * 2 / 1e6, stream / 1e9, stream / best / 1e9, int((dp > 0).sum()), int((al > 0).sum()), int(dp.sum())))
