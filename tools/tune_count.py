"""Development aid: count-stage kernel times (HIP events inside the library) for the C2 workload under the current
environment's LSG_* tuning knobs."""
import sys, time
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
model = synth.named("C2", n_reads=n_reads)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
best = None
for i in range(4):
    eng.pileup_count()
    s = eng.count_stats()
    cur = (s.ms_total, s.ms_bin, s.ms_walk, s.ms_wave)
    best = cur if best is None or cur[0] < best[0] else best
print("total %.2f bin %.2f walk %.2f wave %.2f" % best, "rows by kernel", list(s.rows_by_kernel), "events by kernel", list(s.events_by_kernel), flush=True)
