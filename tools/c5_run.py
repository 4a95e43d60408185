"""Configuration C5 of BASELINE.json on one GPU: the two-pass cell-type re-annotation loop over RESIDENT reads (SURVEY.md §8e "Config 5",
§8f rows 1-2) at the C2 workload's full size.  Pass 1 = count + call under the automated annotation; target sites = the PASS candidates
of pass 1 (what HighConfidenceCancerVariants keeps is a subset of them); per-cell genotyping of every target on the device
(lsg_genotype_cells); re-annotation = a cell is Cancer iff it is covered at >= 3 targets and mutated at >= 25 % of them
(CellTypeReannotation.py:6-65; cells below coverage leave the table); pass 2 = lsg_set_barcodes with the new table + count + call on
the same resident reads.  Prints the time of every stage; the file-level loop (HCCV filter, tables, steps 2-3) is
pipeline.run_reannotation, tested in tests/test_reanno_pipeline_gpu.py.   usage: python tools/c5_run.py [n_reads]"""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from longsom_amd import synth
from longsom_amd._lib import CallParams, CountParams, GenotypeParams
from longsom_amd.engine import Engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = synth.named("C2", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2); eng.synth_reads(m)
cp = CountParams.longsom_defaults()
kp1 = CallParams.longsom_defaults(min_ac_cells=2, min_ac_reads=3)
res = {"workload": "C2 model at %d reads x %d barcodes, reads resident across both passes" % (n, m.n_cb)}


def timed(f, reps=3):
    f(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    eng.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


(_, ms) = timed(lambda: (eng.pileup_count(cp), eng.call_step1(kp1)))
res["pass1_count_call_ms"] = round(ms, 2)
calls = eng.fetch_calls(candidates_only=True)
is_pass = (calls["site_filter"] == (1 << 31)) & (calls["ct_filter"] == 6).any(axis=1)
tgt = calls[is_pass]
if len(tgt) < 50:                                        # few clean somatic calls on purely synthetic noise: take the strongest candidates
    tgt = calls[np.argsort(-calls["alt_bc"][:, 0, 0].astype(np.int64))[:2000]]
    tgt = tgt[np.argsort(tgt["key"])]
keys = np.ascontiguousarray(tgt["key"]); alt_sym = np.ascontiguousarray(np.where(tgt["n_alt"][:, 0] > 0, tgt["alt"][:, 0, 0], tgt["alt"][:, 1, 0]).astype(np.uint8))
(dp_alt, ms) = timed(lambda: eng.genotype_cells(keys, alt_sym, GenotypeParams.longsom_defaults()))
dp, alt = dp_alt
res["targets"] = int(len(keys)); res["genotype_ms"] = round(ms, 2)
covered = (dp > 0).sum(axis=0); mutated = (alt > 0).sum(axis=0)
keep = covered >= 3
cancer = keep & (mutated >= 0.25 * np.maximum(covered, 1))
table2 = np.where(keep, np.where(cancer, 0, 1), 255).astype(np.uint8)
res["cells_kept"] = int(keep.sum()); res["cells_cancer"] = int(cancer.sum()); res["cells_moved"] = int(((table2 != m.celltype_of) & keep).sum())
kp2 = CallParams.longsom_defaults()


def pass2():
    eng.set_barcodes(table2, 2)
    return eng.pileup_count(cp), eng.call_step1(kp2)


(out2, ms) = timed(pass2)
res["pass2_setbarcodes_count_call_ms"] = round(ms, 2)
res["pass2_rows"] = out2[0][0]; res["pass2_sites_candidates"] = list(out2[1])
# the same table counted from scratch (fresh handle state) gives the same rows: the swap is complete
rows_a = [eng.fetch_counts(ct) for ct in range(2)]
eng.set_barcodes(m.celltype_of, 2); eng.synth_reads(m)          # the generator plants its somatic variants by the table that is set: regenerate the SAME reads
eng.set_barcodes(table2, 2); eng.pileup_count(cp)
rows_b = [eng.fetch_counts(ct) for ct in range(2)]
res["pass2_equals_fresh_count"] = all(np.array_equal(a[i], b[i]) for a, b in zip(rows_a, rows_b) for i in range(3))
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c5_run.json", "w"), indent=1)
