"""GPU: the fused two-pass re-annotation run (one decode, reads resident, barcode table swapped for pass 2) writes the same
files as the reference's rule graph executed script by script with the drop-in shims (each pass decoding the BAM again)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from longsom_amd import hostio, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "workflow", "scripts_gpu")


def run(script, *args):
    subprocess.check_call([sys.executable, os.path.join(S, script)] + [str(a) for a in args], cwd=ROOT)


def strip_date(path):
    return "\n".join(l for l in open(path).read().split("\n") if not l.startswith("##fileDate="))


def test_two_pass_loop(engine, tmp_path):
    m = synth.named("C1", n_reads=30000, n_genes=12, n_cb=80, snp_mod=120)
    bam, fa, bct = str(tmp_path / "S1.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa)
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    rp = pipeline.ReannoParams(chain=pipeline.SnvParams(min_ac_cells=2, min_ac_reads=3), hccv_min_depth=10, hccv_delta_vaf=0.05, hccv_delta_mcf=0.05,
                               hccv_clust_dist=5, chrm_contaminant="True", min_variants=2, min_fraction=0.2)
    sp = pipeline.SnvParams()
    fused = tmp_path / "fused"
    out = pipeline.run_reannotation(bam, bct, fa, str(fused), "S1", rp, sp, engine=engine, pass1_step3=True)
    n_hccv = sum(1 for l in open(out.hccv) if not l.startswith("#"))
    assert n_hccv >= 3, "the synthetic sample must yield HCCVs for this test to mean anything"
    assert out.pass2 is not None and 0 < out.n_cancer < out.n_cells_kept <= 80

    # the same graph, rule by rule (every script decodes its own input files)
    w = tmp_path / "chain"
    c = rp.chain

    def chain(sub, barcodes, p):
        d = w / sub
        for x in ("SplitBam", "BaseCellCounter/S1", "MergeCounts", "BaseCellCalling"):
            os.makedirs(d / x, exist_ok=True)
        run("PreProcessing/SplitBamCellTypes.py", "--bam", bam, "--meta", barcodes, "--id", "S1", "--outdir", d / "SplitBam", "--min_MQ", p.min_mapping_quality)
        for ct in ("Cancer", "Non-Cancer"):
            run("SNVCalling/BaseCellCounter.py", "--bam", d / "SplitBam" / ("S1.%s.bam" % ct), "--ref", fa, "--chrom", "all", "--out_folder",
                d / "BaseCellCounter" / "S1", "--min_mq", p.min_mapping_quality, "--tmp_dir", d / "BaseCellCounter" / "S1" / ("temp_" + ct))
        merged = d / "MergeCounts" / "S1.BaseCellCounts.AllCellTypes.tsv"
        run("SNVCalling/MergeBaseCellCounts.py", "--tsv_folder", d / "BaseCellCounter" / "S1", "--outfile", merged)
        pre = d / "BaseCellCalling" / "S1"
        run("SNVCalling/BaseCellCalling.step1.py", "--infile", merged, "--ref", fa, "--outfile", pre, "--min_cell_types", p.min_cell_types, "--min_ac_reads",
            p.min_ac_reads, "--min_ac_cells", p.min_ac_cells, "--alpha1", p.alpha1, "--beta1", p.beta1, "--alpha2", p.alpha2, "--beta2", p.beta2)
        run("SNVCalling/BaseCellCalling.step2.py", "--infile", str(pre) + ".calling.step1.tsv", "--outfile", pre, "--editing", "/nonexistent", "--pon_SR",
            "/nonexistent", "--pon_LR", "--gnomAD_db", "/nonexistent", "--allow_missing_gnomad", "--gnomAD_max", p.max_gnomad_vaf, "--min_distance", p.min_distance)
        run("SNVCalling/BaseCellCalling.step3.py", "--infile", str(pre) + ".calling.step2.tsv", "--outfile", pre, "--chrM_contaminant", "True", "--deltaVAF",
            p.delta_vaf, "--deltaMCF", p.delta_mcf, "--min_ac_reads", p.min_ac_reads, "--min_ac_cells", p.min_ac_cells, "--clust_dist", p.clust_dist)
        return d

    d1 = chain("CellTypeReannotation", bct, c)
    os.makedirs(d1 / "HCCV", exist_ok=True); os.makedirs(d1 / "ReannotatedCellTypes", exist_ok=True)
    run("CellTypeReannotation/HighConfidenceCancerVariants.py", "--SNVs", d1 / "BaseCellCalling" / "S1.calling.step2.tsv", "--outfile", d1 / "HCCV" / "S1",
        "--min_dp", rp.hccv_min_depth, "--deltaVAF", rp.hccv_delta_vaf, "--deltaMCF", rp.hccv_delta_mcf, "--clust_dist", rp.hccv_clust_dist)
    run("CellTypeReannotation/HCCVSingleCellGenotype.py", "--bam", bam, "--infile", d1 / "HCCV" / "S1.HCCV.tsv", "--ref", fa, "--outfile",
        d1 / "HCCV" / "S1.SNVs.SingleCellGenotype.tsv", "--meta", bct, "--alt_flag", rp.alt_flag, "--min_mq", c.min_mapping_quality, "--pvalue", rp.pvalue,
        "--alpha2", c.alpha2, "--beta2", c.beta2, "--chrM_contaminant", rp.chrm_contaminant, "--tmp_dir", d1 / "HCCV" / "S1")
    fus = tmp_path / "nofusions.tsv"
    fus.write_text("#FusionName\tBC\n")
    run("CellTypeReannotation/CellTypeReannotation.py", "--SNVs", d1 / "HCCV" / "S1.SNVs.SingleCellGenotype.tsv", "--fusions", fus, "--outfile",
        d1 / "ReannotatedCellTypes" / "S1.tsv", "--meta", bct, "--min_variants", rp.min_variants, "--min_frac", rp.min_fraction)
    chain("SNVCalling", d1 / "ReannotatedCellTypes" / "S1.tsv", sp)

    rels = ["CellTypeReannotation/HCCV/S1.HCCV.tsv", "CellTypeReannotation/HCCV/S1.SNVs.SingleCellGenotype.tsv",
            "CellTypeReannotation/ReannotatedCellTypes/S1.tsv"]
    for sub in ("CellTypeReannotation", "SNVCalling"):
        rels += [sub + "/" + r for r in ("BaseCellCounter/S1/S1.Cancer.tsv", "BaseCellCounter/S1/S1.Non-Cancer.tsv", "MergeCounts/S1.BaseCellCounts.AllCellTypes.tsv",
                                         "BaseCellCalling/S1.calling.step1.tsv", "BaseCellCalling/S1.calling.step2.tsv", "BaseCellCalling/S1.calling.step3.tsv")]
    for rel in rels:
        assert strip_date(str(fused / rel)) == strip_date(str(w / rel)), rel
    for sub in ("CellTypeReannotation", "SNVCalling"):            # SplitBam report counters (the time column differs)
        a = open(fused / sub / "SplitBam" / "S1.report.txt").read().split("\n")
        b = open(w / sub / "SplitBam" / "S1.report.txt").read().split("\n")
        assert a[0] == b[0] and a[1].split("\t")[:-1] == b[1].split("\t")[:-1], sub
    # pass 2 really ran under another table
    assert open(fused / "SNVCalling/MergeCounts/S1.BaseCellCounts.AllCellTypes.tsv").read() != open(fused / "CellTypeReannotation/MergeCounts/S1.BaseCellCounts.AllCellTypes.tsv").read()


@pytest.mark.parametrize("world,indexed", [(2, False), (3, True)])
def test_two_pass_loop_over_ranks(tmp_path, world, indexed):
    """BASELINE config 5 over several ranks (torch.distributed.run, both ranks on device 0 with the collectives over gloo — RCCL
    refuses two ranks on one GPU; on the 8-GPU node the same code runs with LSG_DIST_BACKEND unset = nccl): every rank keeps its
    region's reads resident across both passes, the HCCV sites are genotyped where their reads are, and every file of both
    passes is the single-process fused run's, byte for byte"""
    import json
    import socket
    m = synth.named("C1", n_reads=30000, n_genes=12, n_cb=80, snp_mod=120)
    bam, fa, bct = str(tmp_path / "S1.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "barcodes.tsv")
    hostio.synth_bam(m, bam, fa)
    if indexed:
        hostio.build_bai(bam)                      # every rank ingests only its slice of the file
    hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
    flags = ["--p1_min_ac_cells", "2", "--p1_min_ac_reads", "3", "--reanno_hccv_min_depth", "10", "--reanno_hccv_delta_vaf", "0.05", "--reanno_hccv_delta_mcf", "0.05",
             "--reanno_hccv_clust_dist", "5", "--reanno_chrm_contaminant", "True", "--reanno_min_variants", "2", "--reanno_min_fraction", "0.2", "--pass1_step3"]
    script = os.path.join(S, "CellTypeReannotation", "longsom_gpu_reannotation.py")
    base = ["--bam", bam, "--meta", bct, "--ref", fa, "--id", "S1"] + flags
    one = tmp_path / "one"
    r1 = subprocess.run([sys.executable, script] + base + ["--outdir", str(one)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    s1 = json.loads([l for l in r1.stdout.split("\n") if l.startswith("{")][-1])
    assert s1["pass2_step3"] and 0 < s1["cancer_cells"] < s1["cells_kept"]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    many = tmp_path / "ranks"
    rn = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                         "--master-port", str(port), script] + base + ["--outdir", str(many)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert rn.returncode == 0, rn.stderr[-3000:]
    sn = json.loads([l for l in rn.stdout.split("\n") if l.startswith("{")][-1])
    assert sn["ranks"] == world and (sn["cells_kept"], sn["cancer_cells"]) == (s1["cells_kept"], s1["cancer_cells"])
    rels = ["CellTypeReannotation/HCCV/S1.HCCV.tsv", "CellTypeReannotation/HCCV/S1.SNVs.SingleCellGenotype.tsv", "CellTypeReannotation/ReannotatedCellTypes/S1.tsv"]
    for sub in ("CellTypeReannotation", "SNVCalling"):
        rels += [sub + "/" + r for r in ("BaseCellCounter/S1/S1.Cancer.tsv", "BaseCellCounter/S1/S1.Non-Cancer.tsv", "MergeCounts/S1.BaseCellCounts.AllCellTypes.tsv",
                                         "BaseCellCalling/S1.calling.step1.tsv", "BaseCellCalling/S1.calling.step2.tsv", "BaseCellCalling/S1.calling.step3.tsv",
                                         "BaseCellCalling/S1.calling.step3.unfiltered.tsv")]
    for rel in rels:
        assert strip_date(str(many / rel)) == strip_date(str(one / rel)), rel
    for sub in ("CellTypeReannotation", "SNVCalling"):
        a = open(many / sub / "SplitBam" / "S1.report.txt").read().split("\n")
        b = open(one / sub / "SplitBam" / "S1.report.txt").read().split("\n")
        assert a[0] == b[0] and a[1].split("\t")[:-1] == b[1].split("\t")[:-1], sub
