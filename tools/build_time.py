"""Development aid (GPU box): wall time of one load (lsg_load_reads: the tile store is built from compact device arrays) on the C2
workload, cold (first load of the process: device allocations included) and warm (buffers reused), with the build's phases."""
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
reads = eng.synth_generate(model); torch.cuda.synchronize()
for label in ("cold", "warm", "warm"):
    t0 = time.perf_counter(); eng.load_reads_struct(reads); torch.cuda.synchronize()
    print("%s load: %.1f ms wall, phases %s" % (label, (time.perf_counter() - t0) * 1e3, eng.build_times()), flush=True)
