"""End-to-end timing of the fused SNV run on a synthetic C2-shaped BAM (files in, files out): where the wall-clock goes once the
kernels are fast.  Writes gpurun_out/end_to_end.json (copied to profiles/rNN_end_to_end.json, which bench.py quotes in config.end_to_end).
usage: python tools/e2e_perf.py [n_reads] [window_gb]      window_gb > 0: the streamed / windowed form (decode overlaps the GPU work)"""
import json, os, shutil, sys, tempfile, threading, time
sys.path.insert(0, ".")


def _heartbeat():                                  # (a 10 M-read BAM takes minutes to write; a silent run looks hung to the GPU box's watchdog)
    t0 = time.time()
    while True:
        time.sleep(60)
        print("... %.0f s" % (time.time() - t0), flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
from longsom_amd import hostio, pipeline, synth
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
window_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
m = synth.named("C2", n_reads=n_reads)
d = tempfile.mkdtemp(prefix="lsg_e2e_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
t0 = time.time(); hostio.synth_bam(m, bam, fa); t_bam = time.time() - t0
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
t_bai = None
if window_gb > 0:                                  # the streamed run takes its windows through the index (device ingest of every window's slice)
    t0 = time.time(); hostio.build_bai(bam); t_bai = time.time() - t0
res = {"workload": "C2 model at %d reads x %d barcodes, written as a BAM (%.0f MB) + FASTA + barcodes.tsv" % (n_reads, m.n_cb, os.path.getsize(bam) / 1e6),
       "host_threads": os.cpu_count(), "runs": {}}
sz = lambda p: os.path.getsize(p) / 1e6
for name, kw in (("whole", {}),) + ((("windowed_%.2fGiB" % window_gb, {"window_bytes": int(window_gb * (1 << 30))}),) if window_gb > 0 else ()):
    out_dir = os.path.join(d, "out_" + name)
    t0 = time.time()
    out = pipeline.run_snv(bam, bct, fa, out_dir, "S", **kw)
    wall = time.time() - t0
    res["runs"][name] = {"wall_s": round(wall, 2), "seconds": {k: round(v, 3) for k, v in out.timings.items()},
                         "out_MB": {"counts": round(sum(sz(p) for p in out.counts.values()), 1), "merged": round(sz(out.merged), 1), "step1": round(sz(out.step1), 1)},
                         "step3_rows": sum(1 for l in open(out.step3) if not l.startswith("#")) - 1}
    print(name, json.dumps(res["runs"][name]), flush=True)
    shutil.rmtree(out_dir, ignore_errors=True)
res["bam_write_s"] = round(t_bam, 1)
if t_bai is not None:
    res["bai_write_s"] = round(t_bai, 1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/end_to_end.json", "w"), indent=1)
shutil.rmtree(d, ignore_errors=True)
