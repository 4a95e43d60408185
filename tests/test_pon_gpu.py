"""GPU: the fused panel-of-normals run (rules/PoN.smk SplitBam_PoN .. PoN in one process, panel from the call records) writes the
same panel as PoN.py's aggregation over the step-1 TABLES the chain wrote, with and without writing those tables."""
import os

import pytest

from longsom_amd import hostio, pipeline, pon, synth

pytestmark = pytest.mark.gpu


def body(path):
    return [l for l in open(path).read().split("\n") if l and not l.startswith("##fileDate=")]


def test_fused_panel_equals_the_table_route(engine, tmp_path):
    normals = []
    for i, (seed_reads, snp_mod) in enumerate(((9000, 90), (7000, 120), (8000, 90))):
        m = synth.named("C1", n_reads=seed_reads, n_genes=6, n_cb=40, snp_mod=snp_mod)
        d = tmp_path / ("n%d" % i)
        os.makedirs(d)
        bam, fa, bct = str(d / "N.bam"), str(d / "ref.fa"), str(d / "barcodes.tsv")      # same genome for every normal (same seed and genes)
        hostio.synth_bam(m, bam, fa)
        hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Epithelial", "Stromal"])
        normals.append(("N%d_Norm" % i, bam, bct))
    ref = str(tmp_path / "n0" / "ref.fa")
    out = pipeline.run_pon(normals, ref, str(tmp_path / "fused"), engine=engine)
    assert out.n_sites > 20
    rows = body(out.pon)
    assert rows[2] == "#CHROM\tPOS\tNum_samples\tSample_ids"
    assert any(r.split("\t")[2] == "3" for r in rows[3:]) and any(r.split("\t")[2] == "1" for r in rows[3:])

    # PoN.py's route: the step-1 tables on disk
    lst = tmp_path / "files.txt"
    lst.write_text("".join(out.step1[n[0]] + "\n" for n in normals))
    by_files = tmp_path / "PoN.files.tsv"
    assert pon.build_from_files(str(lst), str(by_files), 1, "No") == out.n_sites
    assert body(by_files) == rows

    # no per-normal tables, min_samples 2, prefix stripped
    lean = pipeline.run_pon(normals, ref, str(tmp_path / "lean"), min_samples=2, rm_prefix="Yes", write_tables=False, engine=engine)
    assert not os.path.exists(os.path.join(str(tmp_path / "lean"), "PoN", "MergeCounts", "N0_Norm.BaseCellCounts.AllCellTypes.tsv"))
    by_files2 = tmp_path / "PoN.files2.tsv"
    pon.build_from_files(str(lst), str(by_files2), 2, "Yes")
    assert body(by_files2) == body(lean.pon)
    assert 0 < lean.n_sites < out.n_sites

    # the rule file's command line (workflow/rules/PoN.gpu.smk): a process of its own, parameters from flags
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tsv = tmp_path / "normals.tsv"
    tsv.write_text("".join("%s\t%s\t%s\n" % n for n in normals))
    p = pipeline.pon_params()
    subprocess.check_call([sys.executable, os.path.join(root, "workflow", "scripts_gpu", "PoN", "longsom_gpu_pon.py"), "--normals", str(tsv), "--ref", ref,
                           "--outdir", str(tmp_path / "cli"), "--alpha1", repr(p.alpha1), "--beta1", repr(p.beta1), "--alpha2", repr(p.alpha2),
                           "--beta2", repr(p.beta2), "--no_tables"], cwd=root, stdout=subprocess.DEVNULL)
    assert body(tmp_path / "cli" / "PoN" / "PoN" / "PoN_LR.tsv") == rows
    # ... and the datamash-free PoN.py with the reference's flags
    subprocess.check_call([sys.executable, os.path.join(root, "workflow", "scripts_gpu", "PoN", "PoN.py"), "--in_tsv", str(lst), "--out_file",
                           str(tmp_path / "PoN.shim.tsv"), "--min_samples", "1", "--rm_prefix", "No"], cwd=root, stdout=subprocess.DEVNULL)
    assert body(tmp_path / "PoN.shim.tsv") == rows

    # ... and over two ranks (the normals spread over the ranks; both on device 0 here, the gather over gloo): the same panel
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, LSG_DIST_BACKEND="gloo", LSG_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "workflow", "scripts_gpu", "PoN", "longsom_gpu_pon.py"), "--normals", str(tsv), "--ref", ref,
                        "--outdir", str(tmp_path / "ranks"), "--alpha1", repr(p.alpha1), "--beta1", repr(p.beta1), "--alpha2", repr(p.alpha2),
                        "--beta2", repr(p.beta2)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert '"ranks": 2' in r.stdout
    assert body(tmp_path / "ranks" / "PoN" / "PoN" / "PoN_LR.tsv") == rows
    for n in normals:                                      # every normal's tables were written by the rank that took it
        assert body(tmp_path / "ranks" / "PoN" / "BaseCellCalling" / (n[0] + ".calling.step1.tsv")) == body(out.step1[n[0]])
