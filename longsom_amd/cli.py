"""Command-line entry points with the flag surface of the reference's rule scripts, so that
a rule's `shell:` line keeps working when only its `script=` path changes.  Each function names the
reference CLI it stands in for.  `snv` is the fused entry (one decode, everything resident in HBM)."""
import argparse
import glob
import json
import os
import time

from . import calling, hostio, pipeline, tsvio
from ._lib import CallParams, CountParams
from .engine import Engine


def split_bam(argv=None):
    """SplitBamCellTypes.py --bam --meta --id --outdir --min_MQ  (SplitBamCellTypes.py:194-204)"""
    ap = argparse.ArgumentParser(description="Split a BAM into one BAM per cell type by CB tag")
    ap.add_argument("--bam", required=True); ap.add_argument("--meta", required=True); ap.add_argument("--id", default="Sample")
    ap.add_argument("--max_nM", type=int, default=None); ap.add_argument("--max_NH", type=int, default=None)
    ap.add_argument("--min_MQ", type=int, default=255); ap.add_argument("--n_trim", type=int, default=0); ap.add_argument("--outdir", default=".")
    a = ap.parse_args(argv)
    if a.max_nM is not None or a.max_NH is not None or a.n_trim:
        raise SystemExit("--max_nM / --max_NH / --n_trim are not used by LongSom's rules and are not implemented")
    t0 = time.time()
    table = hostio.read_barcodes(a.meta)
    outs = [os.path.join(a.outdir, "%s.%s.bam" % (a.id, ct)) for ct in table.celltype_names]
    rep = hostio.split_bam(a.bam, table, outs, a.min_MQ)
    pipeline.write_report(os.path.join(a.outdir, a.id + ".report.txt"), rep, time.time() - t0)


def _read_bed(path):
    out = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) < 3 or line.startswith(("#", "track", "browser")):
            continue
        out.setdefault(f[0], []).append((int(f[1]), int(f[2])))
    return out


def bed_mask(keys, contig_names, contig_len, bed: str, bed_out: str):
    """Which rows (keys = tid << 32 | 0-based position) lie in the regions MakeWindows leaves (BaseCellCounter.py:81-113): the --bed
    intervals merged when at most 1 bp apart (`merge(d=1)`), clipped to [1, contig length) (`intersect` with the (x, 1, len) list: position 0
    of a contig is never visited, with or without a bed), minus the --bed_out intervals (`subtract`); without --bed, whole contigs.
    window_maker then only cuts the regions into pieces.  bedtools' interval arithmetic restated (pybedtools is not in the reference
    tree nor in this image: unpinned, tests/test_cli_cpu.py holds hand-derived cases)."""
    import numpy as np
    keys = np.asarray(keys, np.int64)
    keep = np.zeros(len(keys), bool)
    want = _read_bed(bed) if bed else None
    drop = _read_bed(bed_out) if bed_out else {}
    tid_of, pos = keys >> 32, keys & 0xFFFFFFFF
    for tid, (name, length) in enumerate(zip(contig_names, contig_len)):
        if want is None:
            ivs = [(1, int(length))]
        else:
            merged = []
            for s0, e0 in sorted(want.get(name, [])):
                if merged and s0 - merged[-1][1] <= 1:
                    merged[-1][1] = max(merged[-1][1], e0)
                else:
                    merged.append([s0, e0])
            ivs = [(max(s0, 1), min(e0, int(length))) for s0, e0 in merged if min(e0, int(length)) > max(s0, 1)]
        sel = np.nonzero(tid_of == tid)[0]
        if not len(sel) or not ivs:
            continue
        p = pos[sel]
        starts = np.asarray([x for x, _ in ivs], np.int64); ends = np.asarray([y for _, y in ivs], np.int64)
        j = np.searchsorted(starts, p, side="right") - 1
        inside = (j >= 0) & (p < ends[np.maximum(j, 0)])
        for s0, e0 in drop.get(name, []):
            inside &= ~((p >= s0) & (p < e0))
        keep[sel] = inside
    return keep


def base_cell_counter(argv=None):
    """BaseCellCounter.py --bam --ref --chrom --out_folder --nprocs --min_mq --tmp_dir [...]  (BaseCellCounter.py:323-342).
    --bam is one cell type's BAM: every CB in it is a cell of that type."""
    ap = argparse.ArgumentParser(description="Per-cell-type pileup base counting on the GPU")
    ap.add_argument("--bam", required=True); ap.add_argument("--ref", required=True); ap.add_argument("--chrom", required=True)
    ap.add_argument("--out_folder", default="."); ap.add_argument("--id"); ap.add_argument("--nprocs", type=int, default=1)
    ap.add_argument("--bin", type=int, default=50000); ap.add_argument("--bed", default=""); ap.add_argument("--bed_out", default="")
    ap.add_argument("--min_ac", type=int, default=0); ap.add_argument("--min_af", type=float, default=0); ap.add_argument("--min_dp", type=int, default=5)
    ap.add_argument("--min_cc", type=int, default=5); ap.add_argument("--min_bq", type=int, default=20); ap.add_argument("--min_mq", type=int, default=255)
    ap.add_argument("--tmp_dir", default="."); ap.add_argument("--device", type=int, default=0)
    _add_htslib_flag(ap)
    a = ap.parse_args(argv)
    _apply_htslib_flag(a)
    if a.min_ac:
        raise SystemExit("--min_ac counts alternative alleles over every read of the pileup column, supplementary ones included (BaseCellCounter.py:152-180,221); "
                         "LongSom's rules never pass it and it is not implemented")
    sid = a.id or os.path.basename(a.bam).replace(".bam", "")
    if a.tmp_dir != ".":
        os.makedirs(a.tmp_dir, exist_ok=True)          # the rule declares it as an output (R:SNVCalling.smk:36)
    dec = hostio.decode_bam(a.bam, None, min_mapq=a.min_mq)
    names, seqs = tsvio.read_fasta(a.ref)
    seq_of = dict(zip(names, seqs))
    with Engine(a.device) as eng:
        eng.set_contigs(dec.contig_len)
        for tid, n in enumerate(dec.contig_names):
            eng.load_reference(tid, seq_of[n])
        import numpy as np
        eng.set_barcodes(np.zeros(max(1, len(dec.barcodes)), np.uint8), 1)
        if a.chrom != "all":
            tid = dec.contig_names.index(a.chrom)
            eng.set_region(tid, 0, tid + 1, 0)
        eng.set_pileup_window(max(64, a.bin))             # --bin: the windows whose pileups each have a max_depth buffer of their own (:185-191)
        cp = CountParams.longsom_defaults(min_bq=a.min_bq, min_mq=a.min_mq, min_dp=a.min_dp, min_cc=a.min_cc)
        eng.set_count_at_load(cp)                         # this script counts its BAM once: in the pass that loads it,
        eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)  # and keeps no store for another count
        eng.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)      # (reads the count would refuse are not stored: keys alone through the sort)
        eng.load_reads(dec.records)
        eng.pileup_count(cp)
        out = os.path.join(a.out_folder, sid + ".tsv")
        print("Outfile: ", out, "\n")
        if not (a.bed or a.bed_out) and os.environ.get("LONGSOM_HOST_TABLES", "0") != "1":
            # the table is printed where its rows lie and streamed into the file (csrc/tables.hip; the same bytes as the writer below)
            eng.set_table_names(dec.contig_names, [sid])
            eng.format_table(eng.TABLE_COUNTS)
            with open(out, "w") as f:
                f.write(tsvio.counts_header(sid))
            eng.append_table(eng.TABLE_COUNTS, out)
            return
        k, r, c = eng.fetch_counts(0)
    if a.bed or a.bed_out:                                # MakeWindows' interval arithmetic (BaseCellCounter.py:88-106): only rows inside are written
        keep = bed_mask(k, dec.contig_names, [len(seq_of[n]) for n in dec.contig_names], a.bed, a.bed_out)
        k, r, c = k[keep], r[keep], c[keep]
    tsvio.write_counts_tsv(out, k, r, c, dec.contig_names, sid)


def merge_counts(argv=None):
    """MergeBaseCellCounts.py --tsv_folder --outfile  (MergeBaseCellCounts.py:206-210); columns in sorted file order."""
    ap = argparse.ArgumentParser(description="Merge per-cell-type base count TSVs")
    ap.add_argument("--tsv_folder", required=True); ap.add_argument("--outfile", required=True)
    a = ap.parse_args(argv)
    files = sorted(glob.glob(a.tsv_folder + "/*.tsv"))
    if not files:
        raise RuntimeError("No tsv files found")
    contigs = tsvio.contigs_of_tsv(files)
    per_ct = [tsvio.parse_counts_tsv(f, contigs)[:3] for f in files]
    cts = [os.path.basename(f).split(".")[-2] for f in files]
    tsvio.write_merged_tsv(a.outfile, per_ct, contigs, cts)


def calling_step1(argv=None):
    """BaseCellCalling.step1.py --infile --outfile --ref --min_cell_types --min_ac_reads --min_ac_cells --alpha1 ...  (step1.py:585-604)"""
    ap = argparse.ArgumentParser(description="Beta-binomial somatic test on the GPU")
    ap.add_argument("--infile", required=True); ap.add_argument("--outfile", required=True); ap.add_argument("--ref", required=True)
    ap.add_argument("--editing"); ap.add_argument("--pon")
    ap.add_argument("--min_cov", type=int, default=5); ap.add_argument("--min_cells", type=int, default=5)
    ap.add_argument("--min_ac_cells", type=int, default=2); ap.add_argument("--min_ac_reads", type=int, default=3)
    ap.add_argument("--max_cell_types", type=int, default=1); ap.add_argument("--min_cell_types", type=int, default=2)
    ap.add_argument("--fisher_cutoff", type=float, default=1); ap.add_argument("--min_distance", type=int, default=5)
    ap.add_argument("--alpha1", type=float, default=0.21356677091082193); ap.add_argument("--beta1", type=float, default=104.95163748636298)
    ap.add_argument("--alpha2", type=float, default=0.2474528917555431); ap.add_argument("--beta2", type=float, default=162.03696139428595)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.fisher_cutoff != 1:
        raise SystemExit("--fisher_cutoff other than 1 (strand test off) is not implemented")
    names, seqs = tsvio.read_fasta(a.ref)
    cts, per_ct, header = tsvio.parse_merged_tsv(a.infile, names)
    with Engine(a.device) as eng:
        eng.set_contigs([len(s) for s in seqs])
        for t, s in enumerate(seqs):
            eng.load_reference(t, s)
        eng.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
        eng.call_step1(CallParams.longsom_defaults(alpha1=a.alpha1, beta1=a.beta1, alpha2=a.alpha2, beta2=a.beta2, min_cov=a.min_cov,
                                                   min_cells=a.min_cells, min_ac_cells=a.min_ac_cells, min_ac_reads=a.min_ac_reads,
                                                   max_cell_types=a.max_cell_types, min_cell_types=a.min_cell_types))
        if os.environ.get("LONGSOM_HOST_TABLES", "0") != "1" and not any(ch in n for n in list(names) + list(cts) for ch in "\t\n"):
            eng.set_table_names(names, cts)
            eng.format_table(eng.TABLE_STEP1)
            with open(a.outfile + ".calling.step1.tsv", "w") as f:
                f.write(tsvio.step1_header(header, cts))
            eng.append_table(eng.TABLE_STEP1, a.outfile + ".calling.step1.tsv")
            return
        calls = eng.fetch_calls()
    tsvio.write_step1_tsv(a.outfile + ".calling.step1.tsv", calls, per_ct, names, cts, header)


def calling_step2(argv=None):
    """BaseCellCalling.step2.py --infile --outfile --editing --pon_SR --pon_LR --gnomAD_db --gnomAD_max --min_distance  (step2.py:237-248).
    --gnomAD_db: the gnomad_db directory the reference's rule passes (read with sqlite3) or a JSON {"chrom:pos:ref:alt": AF};
    a source that cannot be used switches the filter off WITH a warning on stderr."""
    ap = argparse.ArgumentParser(description="Position-set / distance / gnomAD tags")
    ap.add_argument("--infile", required=True); ap.add_argument("--outfile", required=True); ap.add_argument("--editing")
    ap.add_argument("--pon_SR", required=True); ap.add_argument("--pon_LR", nargs="?", const="", default="")
    ap.add_argument("--min_distance", type=int, default=5); ap.add_argument("--gnomAD_db"); ap.add_argument("--gnomAD_max", type=float, default=0.01)
    ap.add_argument("--reference-gz-compat", action="store_true"); ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--allow_missing_gnomad", action="store_true", help="run without the gnomAD filter when --gnomAD_db cannot be read (default: stop, as the reference does)")
    a = ap.parse_args(argv)
    text = open(a.infile, "rb").read()
    contigs = tsvio.contigs_of_tsv([a.infile])
    af = calling.open_gnomad(a.gnomAD_db, a.allow_missing_gnomad or None)               # JSON table, gnomad_db directory / sqlite
    keys = [calling.read_posset_keys(p, contigs, a.reference_gz_compat) for p in (a.editing, a.pon_SR, a.pon_LR)]
    with Engine(a.device) as eng:
        out = calling.step2_bytes(text, eng, contigs, keys[0], keys[1], keys[2], a.min_distance, af, a.gnomAD_max)
    with open(a.outfile + ".calling.step2.tsv", "wb") as f:
        f.write(out)


def calling_step3(argv=None):
    """BaseCellCalling.step3.py --infile --outfile --deltaVAF --deltaMCF --chrM_contaminant --min_ac_reads --min_ac_cells --clust_dist  (step3.py:318-328)"""
    ap = argparse.ArgumentParser(description="LongSom final filters")
    ap.add_argument("--infile", required=True); ap.add_argument("--outfile", required=True)
    ap.add_argument("--deltaVAF", type=float, required=True); ap.add_argument("--deltaMCF", type=float, required=True)
    ap.add_argument("--chrM_contaminant", default="True"); ap.add_argument("--min_ac_reads", type=int, default=2)
    ap.add_argument("--min_ac_cells", type=int, default=3); ap.add_argument("--clust_dist", type=int, default=5)
    a = ap.parse_args(argv)
    final, unfiltered = calling.step3(open(a.infile, "rb").read(), a.deltaVAF, a.deltaMCF, a.min_ac_reads, a.min_ac_cells, a.clust_dist)
    open(a.outfile + ".calling.step3.tsv", "w").write(final)
    open(a.outfile + ".calling.step3.unfiltered.tsv", "w").write(unfiltered)


def hccv(argv=None):
    """HighConfidenceCancerVariants.py --SNVs --outfile PREFIX --min_dp --deltaVAF --deltaMCF [--clust_dist]  (:259-267)."""
    from . import reanno
    ap = argparse.ArgumentParser(description="High-confidence cancer variants for the cell-type re-annotation")
    ap.add_argument("--SNVs", required=True); ap.add_argument("--outfile", required=True)
    ap.add_argument("--min_dp", type=float, required=True); ap.add_argument("--deltaVAF", type=float, required=True)
    ap.add_argument("--deltaMCF", type=float, required=True); ap.add_argument("--clust_dist", type=int, default=10000)
    a = ap.parse_args(argv)
    print("\n- High Confidence Cancer Variants calling\n")
    reanno.hccv_filter(a.SNVs, a.outfile, a.min_dp, a.deltaVAF, a.deltaMCF, a.clust_dist)


def single_cell_genotype(argv=None):
    """HCCVSingleCellGenotype.py --bam --infile --ref --meta --outfile [--alt_flag --nprocs --bin --min_bq --min_mq --tissue
    --tmp_dir --alpha2 --beta2 --pvalue --chrM_contaminant]  (:317-337).  --ref is accepted and not needed (the reference only
    uses it for a count it never reads, :138-145)."""
    from . import reanno
    ap = argparse.ArgumentParser(description="Alleles observed in every cell at the variant sites, on the GPU")
    ap.add_argument("--bam", required=True); ap.add_argument("--infile", required=True); ap.add_argument("--ref", required=True)
    ap.add_argument("--meta", required=True); ap.add_argument("--outfile", default="Matrix.tsv")
    ap.add_argument("--alt_flag", default="All", choices=["Alt", "All"]); ap.add_argument("--nprocs", type=int, default=1)
    ap.add_argument("--bin", type=int, default=50000); ap.add_argument("--min_bq", type=int, default=30); ap.add_argument("--min_mq", type=int, default=255)
    ap.add_argument("--tissue", default=None); ap.add_argument("--tmp_dir", default="tmpDir")
    ap.add_argument("--alpha2", type=float, default=0.260288007167716); ap.add_argument("--beta2", type=float, default=173.94711910763732)
    ap.add_argument("--pvalue", type=float, default=0.01); ap.add_argument("--chrM_contaminant", default="True")
    ap.add_argument("--device", type=int, default=0)
    _add_htslib_flag(ap)
    a = ap.parse_args(argv)
    _apply_htslib_flag(a)
    if a.tissue is not None:
        raise SystemExit("--tissue is not used by LongSom's rules and is not implemented")
    print("Outfile: ", a.outfile, "\n")
    os.makedirs(a.tmp_dir, exist_ok=True)              # the rule declares it as an output (R:CellTypeReannotation.smk:316)
    table = hostio.read_barcodes(a.meta)
    dec = hostio.decode_bam(a.bam, table.barcodes, min_mapq=0)
    with Engine(a.device) as eng:
        eng.set_contigs(dec.contig_len)
        eng.set_barcodes(table.celltype_of, len(table.celltype_names))
        eng.load_reads(dec.records)
        reanno.single_cell_genotype(eng, a.infile, table, dec.contig_names, a.outfile, alt_flag=a.alt_flag, window=a.bin, min_bq=a.min_bq,
                                    min_mq=a.min_mq, alpha2=a.alpha2, beta2=a.beta2, pvalue=a.pvalue, chrm_contaminant=a.chrM_contaminant)


def celltype_reannotation(argv=None):
    """CellTypeReannotation.py --SNVs --fusions --outfile --meta [--min_variants --min_frac]  (:67-77)."""
    from . import reanno
    ap = argparse.ArgumentParser(description="Cancer / non-cancer re-annotation of the cells from their HCCV genotypes")
    ap.add_argument("--SNVs", required=True); ap.add_argument("--fusions", required=True); ap.add_argument("--outfile", required=True)
    ap.add_argument("--meta", required=True); ap.add_argument("--min_variants", type=int, default=3); ap.add_argument("--min_frac", type=float, default=0.2)
    a = ap.parse_args(argv)
    print("Outfile: ", a.outfile, "\n")
    reanno.celltype_reannotation(a.SNVs, a.fusions, a.meta, a.outfile, a.min_variants, a.min_frac)


def _add_gnomad_flag(ap):
    ap.add_argument("--allow_missing_gnomad", action="store_true", help="run without the gnomAD filter when the named source cannot be read (default: stop, as the reference does)")


def _apply_gnomad_flag(a):
    if getattr(a, "allow_missing_gnomad", False):
        os.environ["LONGSOM_ALLOW_MISSING_GNOMAD"] = "1"


def _add_htslib_flag(ap):
    ap.add_argument("--htslib_legacy_del_merge", action="store_true",
                    help="count the first column of a deletion that is followed by another deletion (CIGAR 1D2D) as 'D', as pysam over htslib <= 1.10 "
                         "does; default: htslib >= 1.11 ('O').  The reference's conda environment pins neither (workflow/envs/SComatic.yaml:9,23)")


def _apply_htslib_flag(a):
    if getattr(a, "htslib_legacy_del_merge", False):
        hostio.set_legacy_del_merge(True)


def _optional_paths(ap, *flags):
    """File options a rule may render with an EMPTY value (`--pon_LR {input.pon_LR}` when Run.PoN is False, the reference's
    default config): like the reference's own `--pon_LR` (step2.py:245, nargs='?'), a bare flag means "no file"."""
    for f in flags:
        ap.add_argument(f, nargs="?", const="", default="")


def _add_dataclass_flags(ap, obj, prefix=""):
    for k, v in vars(obj).items():
        if isinstance(v, bool):
            ap.add_argument("--" + prefix + k, action="store_true")
        elif isinstance(v, (int, float, str)):
            ap.add_argument("--" + prefix + k, type=type(v), default=v)


def snv(argv=None):
    """Fused chain: one process from BAM to calling.step3.tsv (workflow/rules/SNVCalling.gpu.smk)."""
    ap = argparse.ArgumentParser(description="SplitBam -> BaseCellCounter -> MergeCounts -> BaseCellCalling step1-3 on one GPU")
    ap.add_argument("--bam", required=True); ap.add_argument("--meta", required=True); ap.add_argument("--ref", required=True)
    ap.add_argument("--id", required=True); ap.add_argument("--outdir", required=True)
    _optional_paths(ap, "--editing", "--pon_SR", "--pon_LR", "--gnomAD_json", "--gnomAD_db")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--window_gb", type=float, default=0.0, help="stream the BAM in batches of about this many GiB of uncompressed BAM and count "
                    "window by window (for a BAM whose reads do not fit in HBM); 0 = the whole BAM at once")
    d = pipeline.SnvParams()
    _add_dataclass_flags(ap, d)
    _add_htslib_flag(ap); _add_gnomad_flag(ap)
    a = ap.parse_args(argv)
    _apply_htslib_flag(a); _apply_gnomad_flag(a)
    params = pipeline.SnvParams(**{k: getattr(a, k) for k in vars(d)})
    # under torch.distributed.run (WORLD_SIZE > 1): one rank per GPU, regions sharded over the ranks; the process group comes up
    # before anything touches the GPU
    from . import regions
    comm = regions.Comm.from_env()
    try:
        out = pipeline.run_snv(a.bam, a.meta, a.ref, a.outdir, a.id, params, a.editing or None, a.pon_SR or None, a.pon_LR or None,
                               a.gnomAD_json or a.gnomAD_db or None, a.device, comm=comm, window_bytes=int(a.window_gb * (1 << 30)) or None)
        if comm.rank == 0:
            print(json.dumps({"outputs": {k: v for k, v in vars(out).items() if k not in ("timings", "resident", "_pending")}, "seconds": out.timings, "ranks": comm.world}))
    finally:
        comm.close()


def reannotation(argv=None):
    """Fused two-pass loop: pass-1 calling, HCCV, per-cell genotyping, re-annotation, pass-2 calling; the BAM is decoded and loaded
    once (workflow/rules/CellTypeReannotation.gpu.smk)."""
    ap = argparse.ArgumentParser(description="CellTypeReannotation + SNVCalling of LongSom on one GPU, reads resident across both passes")
    ap.add_argument("--bam", required=True); ap.add_argument("--meta", required=True); ap.add_argument("--ref", required=True)
    ap.add_argument("--id", required=True); ap.add_argument("--outdir", required=True)
    _optional_paths(ap, "--fusions", "--editing", "--pon_SR", "--pon_LR", "--gnomAD_json", "--gnomAD_db")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--pass1_step3", action="store_true", help="also write pass 1's calling.step3.tsv (the reference's pass 1 stops at step 2)")
    # config['Reanno'] (pass 1: --reanno_* for the HCCV / re-annotation block, --p1_* for its BaseCellCounter / BaseCellCalling
    # block) and config['SNVCalling'] (pass 2: --p2_*); defaults = config/config.yaml
    rp0, sp0 = pipeline.ReannoParams(), pipeline.SnvParams()
    _add_dataclass_flags(ap, rp0, "reanno_"); _add_dataclass_flags(ap, rp0.chain, "p1_"); _add_dataclass_flags(ap, sp0, "p2_")
    _add_htslib_flag(ap); _add_gnomad_flag(ap)
    a = ap.parse_args(argv)
    _apply_htslib_flag(a); _apply_gnomad_flag(a)
    chain = pipeline.SnvParams(**{k: getattr(a, "p1_" + k) for k in vars(rp0.chain)})
    rp = pipeline.ReannoParams(chain=chain, **{k: getattr(a, "reanno_" + k) for k, v in vars(rp0).items() if k != "chain"})
    sp = pipeline.SnvParams(**{k: getattr(a, "p2_" + k) for k in vars(sp0)})
    # under torch.distributed.run (WORLD_SIZE > 1): one rank per GPU, every rank keeps its region's reads resident across both passes
    from . import regions
    comm = regions.Comm.from_env()
    try:
        out = pipeline.run_reannotation(a.bam, a.meta, a.ref, a.outdir, a.id, rp, sp, fusions_tsv=a.fusions or None, editing=a.editing or None,
                                        pon_sr=a.pon_SR or None, pon_lr=a.pon_LR or None, gnomad_af_json=a.gnomAD_json or a.gnomAD_db or None, device=a.device,
                                        pass1_step3=a.pass1_step3, comm=comm)
        if comm.rank == 0:
            print(json.dumps({"hccv": out.hccv, "genotype": out.genotype, "barcodes": out.barcodes, "cells_kept": out.n_cells_kept, "cancer_cells": out.n_cancer,
                              "pass2_step3": out.pass2.step3 if out.pass2 else None, "seconds": out.timings, "ranks": comm.world}))
    finally:
        comm.close()


def pon(argv=None):
    """PoN.py --in_tsv LIST --out_file OUT [--min_samples 2] [--rm_prefix Yes|No]  (scripts/PoN/PoN.py:10-16)."""
    from . import pon as _pon
    ap = argparse.ArgumentParser(description="Script to build a SComatic Panel Of Normals (PoNs)")
    ap.add_argument("--in_tsv", required=True); ap.add_argument("--out_file", required=True)
    ap.add_argument("--min_samples", type=int, default=2); ap.add_argument("--rm_prefix", choices=["Yes", "No"], default="Yes")
    a = ap.parse_args(argv)
    print("-----------------------------------------------------------\n1. Building Panel Of Normals ...\n"
          "-----------------------------------------------------------\n")
    _pon.build_from_files(a.in_tsv, a.out_file, a.min_samples, a.rm_prefix)


def pon_chain(argv=None):
    """rules/PoN.smk SplitBam_PoN .. PoN fused: --normals is a TSV of id, bam, barcodes.tsv (workflow/rules/PoN.gpu.smk)."""
    ap = argparse.ArgumentParser(description="Panel of normals of LongSom on one GPU: count + step-1 call of every normal, then the PoN table")
    ap.add_argument("--normals", required=True); ap.add_argument("--ref", required=True); ap.add_argument("--outdir", required=True)
    ap.add_argument("--alpha1", type=float, required=True); ap.add_argument("--beta1", type=float, required=True)
    ap.add_argument("--alpha2", type=float, required=True); ap.add_argument("--beta2", type=float, required=True)
    ap.add_argument("--min_ac_cells", type=int, default=1); ap.add_argument("--min_ac_reads", type=int, default=1)
    ap.add_argument("--min_cells", type=int, default=1); ap.add_argument("--min_cell_types", type=int, default=1)
    ap.add_argument("--min_mq", type=int, default=60); ap.add_argument("--min_samples", type=int, default=1)
    ap.add_argument("--rm_prefix", choices=["Yes", "No"], default="No"); ap.add_argument("--no_tables", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    _add_htslib_flag(ap)
    a = ap.parse_args(argv)
    _apply_htslib_flag(a)
    normals = [tuple(l.rstrip("\n").split("\t")[:3]) for l in open(a.normals) if l.strip() and not l.startswith("#")]
    p = pipeline.pon_params(alpha1=a.alpha1, beta1=a.beta1, alpha2=a.alpha2, beta2=a.beta2, min_ac_cells=a.min_ac_cells, min_ac_reads=a.min_ac_reads,
                            min_cells=a.min_cells, min_cell_types=a.min_cell_types, min_mapping_quality=a.min_mq)
    # under torch.distributed.run: the normals are spread over the ranks (one GPU each)
    from . import regions
    comm = regions.Comm.from_env()
    try:
        out = pipeline.run_pon(normals, a.ref, a.outdir, p, a.min_samples, a.rm_prefix, not a.no_tables, device=a.device, comm=comm)
        if comm.rank == 0:
            print(json.dumps({"pon": out.pon, "sites": out.n_sites, "seconds": out.timings, "ranks": comm.world}))
    finally:
        comm.close()
