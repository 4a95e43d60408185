"""Rehearsal of BASELINE config 5 over several ranks on the ONE-GPU box: the fused two-pass re-annotation loop of a synthetic C2-shaped
BAM with its .bai under torch.distributed.run, N ranks on device 0, collectives over gloo — every rank ingests its slice once and
keeps it resident across both passes.  Reports rank 0's stage times of both passes, and whether every file of both passes equals the
one-process run's.  Not a scaling number (the ranks share one GPU and one host).
usage: python tools/c5_ranks_rehearsal.py [n_reads] [ranks]"""
import json, os, shutil, socket, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from longsom_amd import hostio, synth
n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = synth.named("C2", n_reads=n_reads)
d = tempfile.mkdtemp(prefix="lsg_c5r_")
bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
hostio.synth_bam(m, bam, fa)
hostio.build_bai(bam)
hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
script = os.path.join("workflow", "scripts_gpu", "CellTypeReannotation", "longsom_gpu_reannotation.py")
base = ["--bam", bam, "--meta", bct, "--ref", fa, "--id", "S"] + sys.argv[3:]


def last_json(text):
    return json.loads([l for l in text.split("\n") if l.startswith("{")][-1])


t0 = time.time(); r1 = subprocess.run([sys.executable, script] + base + ["--outdir", os.path.join(d, "one")], capture_output=True, text=True); t_one = time.time() - t0
if r1.returncode != 0:
    sys.exit(r1.stderr[-3000:])
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port), script] + base + \
      ["--outdir", os.path.join(d, "ranks")]
t0 = time.time(); rn = subprocess.run(cmd, env=env, capture_output=True, text=True); t_n = time.time() - t0
if rn.returncode != 0:
    sys.exit(rn.stderr[-3000:])
s1, sn = last_json(r1.stdout), last_json(rn.stdout)
rels = ["CellTypeReannotation/HCCV/S.HCCV.tsv", "CellTypeReannotation/HCCV/S.SNVs.SingleCellGenotype.tsv", "CellTypeReannotation/ReannotatedCellTypes/S.tsv"]
for sub in ("CellTypeReannotation", "SNVCalling"):
    rels += [sub + "/" + x for x in ("BaseCellCounter/S/S.Cancer.tsv", "BaseCellCounter/S/S.Non-Cancer.tsv", "MergeCounts/S.BaseCellCounts.AllCellTypes.tsv",
                                     "BaseCellCalling/S.calling.step1.tsv", "BaseCellCalling/S.calling.step2.tsv")]
rels.append("SNVCalling/BaseCellCalling/S.calling.step3.tsv")
strip = lambda p: [l for l in open(p, "rb").read().split(b"\n") if not l.startswith(b"##fileDate=")]
present = [x for x in rels if os.path.exists(os.path.join(d, "one", x))]
same = all(os.path.exists(os.path.join(d, "ranks", x)) and strip(os.path.join(d, "one", x)) == strip(os.path.join(d, "ranks", x)) for x in present)
res = {"workload": "C2 model at %d reads as a BAM (%.0f MB) + .bai; two-pass re-annotation loop" % (n_reads, os.path.getsize(bam) / 1e6), "ranks": world,
       "backend": "gloo, every rank on device 0", "one_process_wall_s": round(t_one, 2), "ranks_wall_s": round(t_n, 2),
       "files_compared": len(present), "files_equal_the_one_process_run": same,
       "cells_kept": sn["cells_kept"], "cancer_cells": sn["cancer_cells"], "one_process": {k: s1[k] for k in ("cells_kept", "cancer_cells", "seconds")},
       "rank0_seconds": sn["seconds"]}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c5_ranks_rehearsal.json", "w"), indent=1)
print(json.dumps(res))
shutil.rmtree(d, ignore_errors=True)
