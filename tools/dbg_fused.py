"""development aid: one small fused load (lsg_set_count_at_load) with a synchronisation and a line on stderr after every stage (LSG_DEBUG_SYNC=1)"""
import os, sys
os.environ.setdefault("LSG_DEBUG_SYNC", "1")
sys.path.insert(0, ".")
import numpy as np
from longsom_amd import synth
from longsom_amd._lib import CountParams
from longsom_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000
m = synth.named("C1", n_reads=n)
p = CountParams.longsom_defaults()
with Engine(0) as eng:
    eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2)
    eng.synth_reads(m)
    ref = eng.pileup_count(p); ref_rows = [eng.fetch_counts(ct) for ct in range(2)]
    print("two-pass:", ref, flush=True)
    eng.set_count_at_load(p)
    eng.synth_reads(m)
    print("path", eng.layout_info()[0], flush=True)
    got = eng.pileup_count(p); rows = [eng.fetch_counts(ct) for ct in range(2)]
    print("fused:", got, flush=True)
    for ct in range(2):
        for j in range(3):
            print(ct, j, np.array_equal(rows[ct][j], ref_rows[ct][j]), flush=True)
    got2 = eng.pileup_count(p); rows2 = [eng.fetch_counts(ct) for ct in range(2)]
    print("recount over the fused store:", got2, all(np.array_equal(rows2[ct][j], ref_rows[ct][j]) for ct in range(2) for j in range(3)), flush=True)
