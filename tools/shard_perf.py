"""Development aid: per-rank count + call time of the sharded C2 job, emulated on ONE GPU (one shard after another, no collective;
the shard's load is outside the timed loop here — bench.py times it).  max over ranks of a given N predicts the re-count part of
bench.py --gpus N minus the all-gather."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from longsom_amd import synth
from longsom_amd._lib import CallParams, CountParams
from longsom_amd.engine import Engine
from longsom_amd.shard import region_shards, sub_model, CALL_BYTES

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
model = synth.named("C2", n_reads=n_reads)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
for world in worlds:
    worst = 0.0
    for rank, (lo, hi, g_lo, g_hi) in enumerate(region_shards(model, world)):
        eng.synth_reads(sub_model(model, g_lo, g_hi) if world > 1 else model)
        eng.set_region(lo[0], lo[1], hi[0], hi[1])
        def step():
            rows, cols = eng.pileup_count(cp)
            ns, nc = eng.call_step1(kp)
            npass = eng.export_calls(2, buf.data_ptr(), buf.numel() // CALL_BYTES)      # as bench.py does: rows straight into the send buffer
            return cols, ns, nc, npass
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            cols, ns, nc, npass = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        st = eng.count_stats()
        worst = max(worst, ms)
        print(f"N={world} rank={rank}: {ms:.2f} ms  reads {eng.reads_shape()[0]} cols {cols} sites {ns} cand {nc} pass {npass} | count ms_total {st.ms_total:.2f} walk {st.ms_walk:.2f}", flush=True)
    print(f"N={world}: slowest rank {worst:.2f} ms", flush=True)
