"""Profile of reanno.hccv_filter on the step-2 table of a C2-shaped sample (the table is made by the fused SNV run first; needs the GPU).
usage: python tools/hccv_perf.py [n_reads]"""
import cProfile, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, ".")
from longsom_amd import hostio, pipeline, reanno, synth
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
m = synth.named("C2", n_reads=n_reads)
d = tempfile.mkdtemp(prefix="lsg_hccv_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
hostio.synth_bam(m, bam, fa)
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
out = pipeline.run_snv(bam, bct, fa, os.path.join(d, "out"), "S")
print("step2: %.1f MB, %d rows" % (os.path.getsize(out.step2) / 1e6, sum(1 for l in open(out.step2) if not l.startswith("#"))))
for mode in ("1", "0"):
    os.environ["LONGSOM_HCCV_ROW_PATH"] = mode
    t0 = time.time(); o = reanno.hccv_filter(out.step2, os.path.join(d, "h" + mode), 50, 0.2, 0.25, 10000)
    print("row-wise" if mode == "1" else "column-wise", "%.2f s" % (time.time() - t0), [sum(1 for _ in open(o + s)) for s in ("", "2", "3")])
cProfile.run("reanno.hccv_filter(out.step2, os.path.join(d, 'p'), 50, 0.2, 0.25, 10000)", os.path.join(d, "prof"))
pstats.Stats(os.path.join(d, "prof")).sort_stats("cumulative").print_stats(22)
shutil.rmtree(d, ignore_errors=True)
