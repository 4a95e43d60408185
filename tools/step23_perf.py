"""Where steps 2 and 3 of a fused run spend their time: every call of the native scanners / writers behind them is timed by a wrapper.
usage: python tools/step23_perf.py [n_reads]      (a C2-shaped BAM of n_reads is written first; needs the GPU)"""
import collections, json, os, shutil, sys, tempfile, time
sys.path.insert(0, ".")
from longsom_amd import calling, hostio, pipeline, synth, tsvio

spent = collections.OrderedDict()


def timed(mod, name):
    fn = getattr(mod, name)

    def wrap(*a, **k):
        t0 = time.time()
        try:
            return fn(*a, **k)
        finally:
            e = spent.setdefault(mod.__name__.split(".")[-1] + "." + name, [0, 0.0]); e[0] += 1; e[1] += time.time() - t0
    setattr(mod, name, wrap)


for m, n in ((tsvio, "scan_rows"), (tsvio, "gather_lines"), (tsvio, "step3_rows"), (tsvio, "write_step1_tsv"), (calling, "step2_bytes"), (calling, "step3_bytes"),
             (calling, "_step3_survivors"), (calling, "read_posset_keys")):
    timed(m, n)
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
m = synth.named("C2", n_reads=n_reads)
d = tempfile.mkdtemp(prefix="lsg_s23_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
hostio.synth_bam(m, bam, fa)
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
t0 = time.time(); out = pipeline.run_snv(bam, bct, fa, os.path.join(d, "out"), "S"); wall = time.time() - t0
res = {"reads": n_reads, "wall_s": round(wall, 2), "seconds": {k: round(float(v), 3) for k, v in out.timings.items()},
       "calls": {k: {"n": v[0], "s": round(v[1], 3)} for k, v in spent.items()},
       "MB": {"step1": round(os.path.getsize(out.step1) / 1e6, 1), "step2": round(os.path.getsize(out.step2) / 1e6, 1),
              "step3_unfiltered": round(os.path.getsize(out.step3_unfiltered) / 1e6, 1)}}
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/step23_perf.json", "w"), indent=1)
shutil.rmtree(d, ignore_errors=True)
