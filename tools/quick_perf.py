"""Ad-hoc timing of the count path on device-generated data (development aid; bench.py is the contract)."""
import sys, time
sys.path.insert(0, ".")
from longsom_amd import synth
from longsom_amd.engine import Engine

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
model = synth.named("C2", n_reads=n_reads)
print(f"model built in {time.time()-t0:.1f}s: {model.n_genes} genes", flush=True)
eng = Engine(0)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
t0 = time.time(); eng.synth_reads(model); print(f"generated in {time.time()-t0:.2f}s shape={eng.reads_shape()}", flush=True)
for i in range(reps):
    t0 = time.time()
    rows, cols = eng.pileup_count()
    dt = time.time() - t0
    s = eng.count_stats()
    E = s.n_events_admitted
    print(f"rep{i}: wall {dt*1e3:.1f} ms  bin {s.ms_bin:.2f} deep {s.ms_deep:.2f} wave {s.ms_wave:.2f} total {s.ms_total:.2f} ms | rows {rows} cols {cols} "
          f"| admitted reads {s.n_reads_admitted} segs {s.n_segs_admitted} events {E} entries {s.n_entries} units {s.n_units} deep {s.n_deep_units} "
          f"| ev wave {s.n_events_wave} deep {s.n_events_deep} | GB/s wave {2*s.n_events_wave/max(s.ms_wave,1e-6)/1e6:.1f} deep {2*s.n_events_deep/max(s.ms_deep,1e-6)/1e6:.1f}", flush=True)
