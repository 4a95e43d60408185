"""Configuration C4 of BASELINE.json (50 M reads x 20 k barcodes) as ONE-SHOT steps on one GPU: the compact read-record arrays resident
(106 GB of events), every step = load (entries binned and sorted, the count made in the same pass, no store kept) + merge + step-1
call, as bench.py times C2.  The first step allocates; the later ones are the figure.  Prints one line per step and a JSON summary
(gpurun_out/c4_oneshot.json).   usage: python tools/c4_oneshot.py [n_reads] [steps]"""
import json, os, sys, time
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd._lib import CallParams, CountParams
from longsom_amd.engine import Engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = synth.named("C4", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
eng.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)
t0 = time.perf_counter(); reads = eng.synth_generate(model); torch.cuda.synchronize()
print("generated in %.1f s: %d reads, %d segments, %d events (%.1f GB of compact events resident)" % (time.perf_counter() - t0, reads.n_reads, reads.n_segs, reads.n_events, reads.n_events * 2 / 1e9), flush=True)
eng.set_region()
eng.set_count_at_load(cp); eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
out = {"workload": "C4: %d reads x %d barcodes, one-shot steps (load + count in one pass, no store kept + merge + step-1 call)" % (model.n_reads, model.n_cb), "steps": []}
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.load_reads_struct(reads)
    t1 = time.perf_counter()
    rows, cols = eng.pileup_count(cp)
    ns, nc = eng.call_step1(kp)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = eng.count_stats(); bt = eng.build_times()
    rec = {"step": i, "ms": round((t2 - t0) * 1e3, 2), "load_and_count_ms": round((t1 - t0) * 1e3, 2), "call_ms": round((t2 - t1) * 1e3, 2), "load_path": eng.layout_info()[0],
           "rows": list(rows), "columns": int(cols), "merged_sites": int(ns), "candidates": int(nc), "phases_ms": [round(float(x), 2) for x in bt], "count_kernel_ms": round(float(st.ms_walk), 2),
           "events_admitted": int(st.n_events_admitted), "entries": int(st.n_entries), "sites_per_s": cols / (t2 - t0)}
    out["steps"].append(rec)
    print(rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c4_oneshot.json", "w"), indent=1)
