"""Rehearsal of the sharded product run on the ONE-GPU box: the fused SNV run of a synthetic C2-shaped BAM with its .bai under
torch.distributed.run, N ranks on device 0, collectives over gloo (RCCL refuses two ranks on one GPU) — what each rank ingests (records
and bytes of its slice of the file), the wall time, and whether every output file equals the single-rank run's.  Not a scaling number
(the ranks share one GPU and one host): a record that the N-rank code path runs end to end at this size.
usage: python tools/shard_rehearsal.py [n_reads] [ranks]"""
import filecmp, json, os, shutil, socket, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from longsom_amd import hostio, pipeline, synth
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = synth.named("C2", n_reads=n_reads)
d = tempfile.mkdtemp(prefix="lsg_shard_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
hostio.synth_bam(m, bam, fa)
hostio.build_bai(bam)
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
t0 = time.time(); one = pipeline.run_snv(bam, bct, fa, os.path.join(d, "one"), "S"); t_one = time.time() - t0
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
       os.path.join("workflow", "scripts_gpu", "SNVCalling", "longsom_gpu_snv.py"), "--bam", bam, "--meta", bct, "--ref", fa, "--id", "S", "--outdir", os.path.join(d, "ranks")]
t0 = time.time(); r = subprocess.run(cmd, env=env, capture_output=True, text=True); t_n = time.time() - t0
if r.returncode != 0:
    sys.exit(r.stderr[-3000:])
summary = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
rels = ["BaseCellCounter/S/S.Cancer.tsv", "BaseCellCounter/S/S.Non-Cancer.tsv", "MergeCounts/S.BaseCellCounts.AllCellTypes.tsv", "BaseCellCalling/S.calling.step1.tsv",
        "BaseCellCalling/S.calling.step2.tsv", "BaseCellCalling/S.calling.step3.tsv", "BaseCellCalling/S.calling.step3.unfiltered.tsv"]
strip = lambda p: [l for l in open(p, "rb").read().split(b"\n") if not l.startswith(b"##fileDate=")]
same = all(strip(os.path.join(d, "one", x)) == strip(os.path.join(d, "ranks", x)) for x in rels)
res = {"workload": "C2 model at %d reads as a BAM (%.0f MB) + .bai" % (n_reads, os.path.getsize(bam) / 1e6), "ranks": world, "backend": "gloo, every rank on device 0",
       "single_rank_wall_s": round(t_one, 2), "ranks_wall_s": round(t_n, 2), "files_equal_the_single_rank_run": same,
       "records_in_file": sum(summary["seconds"]["ingest_records_by_rank"]) if False else None,
       "ingest_records_by_rank": summary["seconds"]["ingest_records_by_rank"], "ingest_slice_MB_by_rank": summary["seconds"]["ingest_slice_MB_by_rank"],
       "rank0_seconds": {k: v for k, v in summary["seconds"].items() if not isinstance(v, list)}}
del res["records_in_file"]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/shard_rehearsal.json", "w"), indent=1)
print(json.dumps(res))
shutil.rmtree(d, ignore_errors=True)
