"""Development aid (GPU box): wall time of the per-load build (tile index + tile-major store) on the C2 workload, cold (first build of
the process: device allocations included) and warm (same reads generated again: buffers reused)."""
import os, sys, time
os.environ["LSG_TIMING"] = "1"
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd._lib import CountParams
from longsom_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
model = synth.named("C2", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
cp = CountParams.longsom_defaults()
for label in ("cold", "warm", "warm"):
    eng.synth_reads(model); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.prepare_counts(cp); torch.cuda.synchronize()
    print("%s build: %.1f ms wall" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
