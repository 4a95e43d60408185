"""Configuration C4 of BASELINE.json (50 M reads x 20 k barcodes) on one GPU: does it fit, how long does a pass take, and do two
size-independent properties hold (region split invariance of the totals; the call stage covers every counted site)."""
import sys, time
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
model = synth.named("C4", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
t0 = time.time(); eng.synth_reads(model); torch.cuda.synchronize(); print("generated + aligned in %.1f s, shape %s, free %.0f GB" % (time.time() - t0, eng.reads_shape(), torch.cuda.mem_get_info()[0] / 1e9), flush=True)
for i in range(2):
    t0 = time.perf_counter(); rows, cols = eng.pileup_count(); ns, nc = eng.call_step1(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = eng.count_stats()
    print("pass %d: %.1f ms  rows %s cols %d sites %d cand %d | events %d entries %d units %d deep %d | bin %.1f walk %.1f wave %.1f" %
          (i, dt * 1e3, rows, cols, ns, nc, s.n_events_admitted, s.n_entries, s.n_units, s.n_deep_units, s.ms_bin, s.ms_walk, s.ms_wave), flush=True)
full = (rows, cols, ns)
# property: counting two halves of the genome separately gives the same totals
tid_mid = len(model.contig_len) // 2
tot_rows, tot_cols, tot_sites = [0, 0], 0, 0
for lo, hi in (((0, 0), (tid_mid, 0)), ((tid_mid, 0), (len(model.contig_len), 0))):
    eng.set_region(lo[0], lo[1], hi[0], hi[1]); r, c = eng.pileup_count(); s_, _ = eng.call_step1()
    tot_rows = [a + b for a, b in zip(tot_rows, r)]; tot_cols += c; tot_sites += s_
print("split totals", tot_rows, tot_cols, tot_sites, "match" if (tot_rows == list(full[0])[:2] and tot_cols == full[1] and tot_sites == full[2]) else "MISMATCH", flush=True)
