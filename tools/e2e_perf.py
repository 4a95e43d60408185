"""End-to-end timing of the fused SNV run on a synthetic BAM (files in, files out): where the wall-clock goes once the
kernels are fast.  usage: python tools/e2e_perf.py [n_reads]"""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
from longsom_amd import hostio, pipeline, synth
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
m = synth.named("C1", n_reads=n_reads, n_genes=400, n_cb=2000, snp_mod=300)
d = tempfile.mkdtemp(prefix="lsg_e2e_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
t0 = time.time(); hostio.synth_bam(m, bam, fa); t_bam = time.time() - t0
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
t0 = time.time()
out = pipeline.run_snv(bam, bct, fa, os.path.join(d, "out"), "S")
wall = time.time() - t0
sz = lambda p: os.path.getsize(p) / 1e6
print(json.dumps({"reads": n_reads, "bam_MB": round(sz(bam), 1), "wall_s": round(wall, 2), "seconds": {k: round(v, 3) for k, v in out.timings.items()},
                  "out_MB": {"counts": round(sum(sz(p) for p in out.counts.values()), 1), "merged": round(sz(out.merged), 1), "step1": round(sz(out.step1), 1)},
                  "bam_write_s": round(t_bam, 1)}))
