"""Does a repeated call stage over the same counted rows, and a repeated count, give the same candidate count?  (C4-sized by default:
the configuration where a race on shared cache lines would show.)  usage: python tools/determinism_check.py [config] [n_reads]"""
import sys
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else (50_000_000 if cfg == "C4" else 10_000_000)
model = synth.named(cfg, n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
eng.synth_reads(model)
for c in range(3):
    rows, cols = eng.pileup_count()
    cands = [eng.call_step1()[1] for _ in range(3)]
    print("count %d: rows %s cols %d -> candidates of three call passes %s" % (c, rows, cols, cands), flush=True)
