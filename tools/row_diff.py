"""Which resident rows differ between two counts of the same reads?  (debug aid for nondeterminism)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
model = synth.named("C4", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
ref = None
for it in range(4):
    eng.pileup_count()
    k, r, c = eng.fetch_counts(0)
    if ref is None:
        ref = (k, c); print("rows", len(k), flush=True); continue
    assert np.array_equal(k, ref[0])
    bad = np.nonzero((c != ref[1]).any(axis=1))[0]
    print("count", it, "rows differing from count 0:", len(bad), flush=True)
    import collections
    colhist = collections.Counter()
    for b in bad:
        for cc in np.nonzero(c[b] != ref[1][b])[0]: colhist[int(cc)] += 1
    print("  columns hit:", sorted(colhist.items()), flush=True)
    for b in bad[:3]:
        print("  row", int(b), "pos", int(k[b]) & 0xffffffff, "\n    now", c[b][:34].tolist(), "\n    ref", ref[1][b][:34].tolist(), flush=True)
