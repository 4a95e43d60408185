"""Hashes of the count rows and the candidate call records of a synthetic sample: two builds of the library must print the same
lines.  usage: python tools/rows_hash.py <package dir> [config] [n_reads]   (package dir = directory holding longsom_amd/)"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import xxhash
from longsom_amd import synth
from longsom_amd.engine import Engine
cfg = sys.argv[2] if len(sys.argv) > 2 else "C4"
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 5_000_000
model = synth.named(cfg, n_reads=n)
eng = Engine(0)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
rows, cols = eng.pileup_count()
ns, nc = eng.call_step1()
print("rows", rows, "cols", cols, "sites", ns, "cand", nc)
for ct in range(2):
    k, r, c = eng.fetch_counts(ct)
    print("ct", ct, "keys", xxhash.xxh64(np.ascontiguousarray(k).tobytes()).hexdigest(), "refs", xxhash.xxh64(np.ascontiguousarray(r).tobytes()).hexdigest(),
          "counts", xxhash.xxh64(np.ascontiguousarray(c).tobytes()).hexdigest())
calls = eng.fetch_calls(candidates_only=True)
print("calls", len(calls), xxhash.xxh64(calls.tobytes()).hexdigest())
