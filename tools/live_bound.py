"""Load-time depth bound (lsg_max_live_reads) of the BASELINE workloads C2 and C4."""
import sys, time
sys.path.insert(0, ".")
from longsom_amd import synth
from longsom_amd.engine import Engine
for cfg, n in (("C2", 10_000_000), ("C4", 50_000_000)):
    model = synth.named(cfg, n_reads=n)
    eng = Engine(0)
    eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
    t0 = time.time(); eng.synth_reads(model); dt = time.time() - t0
    print(cfg, "reads", n, "max_live_reads", eng.max_live_reads(), "gen+layout s", round(dt, 2), flush=True)
    del eng
