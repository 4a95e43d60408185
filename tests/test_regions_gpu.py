"""GPU: the sharded and the windowed forms of the fused SNV run write, byte for byte, the files of the single-pass run.
  * windows: the BAM streamed in small batches, counted window by window with carried reads (a BAM larger than HBM);
  * ranks: two ranks under torch.distributed.run, regions balanced by events, candidate rows all-gathered — rehearsed on the
    one-GPU box with both ranks on device 0 and the collectives over gloo (RCCL refuses two ranks on one GPU; on the 8-GPU node
    the same code path runs with LSG_DIST_BACKEND unset = nccl)."""
import os
import socket
import subprocess
import sys

import pytest

from longsom_amd import hostio, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
RELS = ["BaseCellCounter/S1/S1.Cancer.tsv", "BaseCellCounter/S1/S1.Non-Cancer.tsv", "MergeCounts/S1.BaseCellCounts.AllCellTypes.tsv",
        "BaseCellCalling/S1.calling.step1.tsv", "BaseCellCalling/S1.calling.step2.tsv", "BaseCellCalling/S1.calling.step3.tsv",
        "BaseCellCalling/S1.calling.step3.unfiltered.tsv"]


def strip_date(path):
    return "\n".join(l for l in open(path).read().split("\n") if not l.startswith("##fileDate="))


def same_outputs(a, b):
    for rel in RELS:
        assert strip_date(os.path.join(a, rel)) == strip_date(os.path.join(b, rel)), rel
    ra = open(os.path.join(a, "SplitBam/S1.report.txt")).read().split("\n")
    rb = open(os.path.join(b, "SplitBam/S1.report.txt")).read().split("\n")
    assert ra[0] == rb[0] and ra[1].split("\t")[:-1] == rb[1].split("\t")[:-1]


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("regions")
    m = synth.named("C1", n_reads=24000, n_genes=60, n_cb=90, snp_mod=200)
    bam, fa, bct = str(d / "S1.bam"), str(d / "ref.fa"), str(d / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa)
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    whole = str(d / "whole")
    pipeline.run_snv(bam, bct, fa, whole, "S1")
    assert sum(1 for l in open(os.path.join(whole, RELS[5])) if not l.startswith("#")) > 0
    return bam, fa, bct, whole, d


@pytest.mark.parametrize("window_bytes", [300_000, 4_000_000])
def test_windowed_run_equals_whole_run(sample, window_bytes):
    bam, fa, bct, whole, d = sample
    out_dir = str(d / ("win%d" % window_bytes))
    out = pipeline.run_snv(bam, bct, fa, out_dir, "S1", window_bytes=window_bytes)
    assert out.timings["windows"] >= (20 if window_bytes < 1_000_000 else 2)
    same_outputs(out_dir, whole)
    assert not os.path.exists(os.path.join(out_dir, "_pieces.S1"))


def test_windowed_run_of_the_reference_pinned_sample(tmp_path):
    """the multi-contig fixture (file order chr1, chr10, chr2, chrM): windows == the tables the reference's code wrote"""
    out = pipeline.run_snv(os.path.join(G, "pileup.rand.bam"), os.path.join(G, "pileup.rand.barcodes.tsv"), os.path.join(G, "pileup.rand.fa"),
                           str(tmp_path), "s", window_bytes=50_000)
    assert out.timings["windows"] > 3
    for cname in ("Cancer", "Non-Cancer"):
        got = "".join(l for l in open(out.counts[cname]).read().splitlines(True) if not l.startswith("##fileDate="))
        assert got == open(os.path.join(G, "pileup.rand.%s.tsv" % cname)).read()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharded_run_equals_whole_run(sample, world):
    bam, fa, bct, whole, d = sample
    out_dir = str(d / ("ranks%d" % world))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "workflow", "scripts_gpu", "SNVCalling", "longsom_gpu_snv.py"), "--bam", bam, "--meta", bct, "--ref", fa, "--id", "S1", "--outdir", out_dir]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert '"ranks": %d' % world in r.stdout
    same_outputs(out_dir, whole)


def test_ranks_ingest_only_their_slice_of_an_indexed_bam(sample, tmp_path):
    """with a .bai beside the BAM every rank ingests the slice of the file its region needs (regions.BaiPlan, lsg_load_bam_range) — the
    three ranks together read about the file once plus the reads that reach across the two boundaries, not three times the file —
    SplitBam's counters are summed over the ranks, and every output file is the single-rank run's, byte for byte"""
    import json
    import shutil
    bam0, fa, bct, whole, d = sample
    bam = str(tmp_path / "S1.bam")
    shutil.copy(bam0, bam)
    hostio.build_bai(bam)
    n_records = sum(int(x) for x in open(os.path.join(whole, "SplitBam/S1.report.txt")).read().split("\n")[1].split("\t")[:1])      # Total_reads
    world = 3
    out_dir = str(tmp_path / "ranks")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "workflow", "scripts_gpu", "SNVCalling", "longsom_gpu_snv.py"), "--bam", bam, "--meta", bct, "--ref", fa, "--id", "S1", "--outdir", out_dir]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    by_rank = summary["seconds"]["ingest_records_by_rank"]
    assert len(by_rank) == world and all(x > 0 for x in by_rank)
    assert max(by_rank) < 0.6 * n_records and sum(by_rank) < 1.5 * n_records, (by_rank, n_records)
    assert sum(summary["seconds"]["ingest_slice_MB_by_rank"]) < 1.5 * os.path.getsize(bam) / 1e6
    same_outputs(out_dir, whole)


@pytest.mark.parametrize("window_bytes", [300_000, 4_000_000])
def test_windowed_run_of_an_indexed_bam_ingests_on_the_device(sample, tmp_path, window_bytes):
    """with a .bai the windows are regions of the index and every window's slice of the file is ingested on the GPU (no host decode, no
    carried reads): the files of the single-pass run"""
    import shutil
    bam0, fa, bct, whole, d = sample
    bam = str(tmp_path / "S1.bam")
    shutil.copy(bam0, bam)
    hostio.build_bai(bam)
    out_dir = str(tmp_path / "win")
    out = pipeline.run_snv(bam, bct, fa, out_dir, "S1", window_bytes=window_bytes)
    assert out.timings["windows"] >= (5 if window_bytes < 1_000_000 else 1)
    same_outputs(out_dir, whole)


@pytest.mark.parametrize("distance", [0, 150])
def test_sharded_and_windowed_runs_with_position_sets_and_gnomad(sample, tmp_path, distance):
    """step 2 with an RNA-editing set, a PoN and a gnomAD table: with the distance filter off every rank / window tags its own rows
    on its own GPU and only step 3's survivors travel; with it on, rank 0 tags all kept rows as before — both write the files of the
    single-pass run"""
    import json
    bam, fa, bct, whole, d = sample
    cand = [l.split("\t") for l in open(os.path.join(whole, RELS[3])) if not l.startswith("#") and l.split("\t")[4] != "." and l.split("\t")[5] != "."]
    assert len(cand) > 50
    ed, sr, af = str(tmp_path / "editing.tsv"), str(tmp_path / "pon.tsv"), str(tmp_path / "gnomad.json")
    open(ed, "w").write("#c\tp\n" + "".join("%s\t%s\n" % (c[0], c[1]) for c in cand[::7]))
    open(sr, "w").write("".join("%s\t%s\n" % (c[0], c[1]) for c in cand[3::9]))
    json.dump({"%s:%s:%s:%s" % (c[0], c[1], c[3], c[4].split(",")[0].split("|")[0]): (0.2 if i % 2 else 0.001) for i, c in enumerate(cand[5::11])}, open(af, "w"))
    params = pipeline.SnvParams(min_distance=distance)
    ref_dir = str(tmp_path / "whole")
    pipeline.run_snv(bam, bct, fa, ref_dir, "S1", params, editing=ed, pon_sr=sr, gnomad_af_json=af)
    tagged = open(os.path.join(ref_dir, RELS[4])).read()
    assert "RNA_editing_db" in tagged and "PoN_SR" in tagged and "gnomAD" in tagged and (distance == 0 or "Clustered" in tagged)
    win_dir = str(tmp_path / "win")
    pipeline.run_snv(bam, bct, fa, win_dir, "S1", params, editing=ed, pon_sr=sr, gnomad_af_json=af, window_bytes=500_000)
    same_outputs(win_dir, ref_dir)
    out_dir = str(tmp_path / "ranks")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "workflow", "scripts_gpu", "SNVCalling", "longsom_gpu_snv.py"), "--bam", bam, "--meta", bct, "--ref", fa, "--id", "S1", "--outdir", out_dir,
           "--editing", ed, "--pon_SR", sr, "--gnomAD_json", af, "--min_distance", str(distance)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    same_outputs(out_dir, ref_dir)
