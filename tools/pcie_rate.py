"""PCIe-inclusive rate of the boundary: the C-ABI takes HOST read-record arrays; this times lsg_load_reads
(host -> HBM copy + tile-aligned relayout) on a 1/10 sample of C2 next to one count + call pass over it."""
import sys, time
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
model = synth.named("C2", n_reads=n_reads)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
rec = eng.reads_to_host()
nbytes = sum(getattr(rec, n).nbytes for n, _ in rec._SPEC)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); eng.load_reads(rec); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
t0 = time.perf_counter(); rows, cols = eng.pileup_count(); eng.call_step1(); torch.cuda.synchronize(); t_pass = time.perf_counter() - t0
t0 = time.perf_counter(); rows, cols = eng.pileup_count(); eng.call_step1(); torch.cuda.synchronize(); t_pass = time.perf_counter() - t0
print("reads %d  host arrays %.2f GB  load_reads %.3f s (%.1f GB/s incl. relayout)  count+call %.4f s  sites %d  -> resident %.3e sites/s, PCIe-inclusive %.3e sites/s"
      % (n_reads, nbytes / 1e9, best, nbytes / 1e9 / best, t_pass, cols, cols / t_pass, cols / (t_pass + best)))
