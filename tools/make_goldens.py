#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own Python on small
synthetic inputs.  Runs only in the build container (needs /root/reference);
the outputs are data files (TSV / JSON), never reference source.

What is imported from /root/reference/workflow/scripts (read-only, no bytecode written):
  SNVCalling/MergeBaseCellCounts.py     merge_cell_types_files           (stdlib only)
  SNVCalling/BaseCellCalling.step1.py   variant_calling_step1            (needs `pysam` ONLY for FastaFile.fetch)
  SNVCalling/BaseCellCalling.step2.py   variant_calling_step2            (needs `gnomad_db`)
  SNVCalling/BaseCellCalling.step3.py   variant_calling_step3
Two third-party modules are absent from this image, so minimal stand-ins are registered in
sys.modules before the import (the arithmetic under test — scipy's betabinom, the filter logic,
pandas' text round trips — is the real thing):
  pysam.FastaFile  -> reads the small FASTA we generate; fetch() follows pysam's contract
                      (ValueError for start < 0, clipping at the contig end)
  gnomad_db.database.gnomAD_DB -> get_info_from_df returns AF from a dict we also save as a fixture
                      (gnomAD itself is not in the tree: parity for that column is pinned to this stand-in).
BaseCellCounter.py / SplitBamCellTypes.py cannot be driven this way (their arithmetic IS pysam's
pileup); see tests/golden/kat_pileup.json for the hand-derived known answers instead.
"""
import importlib.util
import json
import os
import shutil
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/workflow/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---- stand-ins for the two absent third-party modules -------------------------------------------
class _FastaFile:
    def __init__(self, path):
        self._seq = {}
        name = None
        with open(path) as f:
            for line in f:
                line = line.rstrip("\n")
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    self._seq[name] = []
                elif name is not None:
                    self._seq[name].append(line)
        self._seq = {k: "".join(v) for k, v in self._seq.items()}
        self.references = list(self._seq)

    def get_reference_length(self, name):
        return len(self._seq[name])

    def fetch(self, reference=None, start=None, end=None):
        if reference not in self._seq:
            raise KeyError(reference)
        if start is not None and start < 0:
            raise ValueError("start out of range (%i)" % start)
        s = self._seq[reference]
        return s[start:end]

    def close(self):
        pass


def install_stubs(gnomad_af):
    pysam = types.ModuleType("pysam")
    pysam.FastaFile = _FastaFile
    sys.modules["pysam"] = pysam
    g = types.ModuleType("gnomad_db")
    gd = types.ModuleType("gnomad_db.database")

    class gnomAD_DB:
        def __init__(self, *a, **k):
            pass

        def get_info_from_df(self, df, col):
            import pandas as pd
            return pd.Series([gnomad_af.get("%s:%s:%s:%s" % (c, p, r, a), float("nan"))
                              for c, p, r, a in zip(df["chrom"], df["pos"], df["ref"], df["alt"])], index=df.index)

    gd.gnomAD_DB = gnomAD_DB
    g.database = gd
    sys.modules["gnomad_db"] = g
    sys.modules["gnomad_db.database"] = gd


def load(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- synthetic inputs -----------------------------------------------------------------------------
HEADER = ("##fileDate=01/01/2000\n"
          '##INFO=DP,Description="Depth of coverage">\n'
          '##INFO=NC,Description="Number of different cells">\n'
          '##INFO=CC,Description="Cell counts [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
          '##INFO=BC,Description="Base counts [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
          '##INFO=BQ,Description="Base quality sums [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
          '##INFO=BCf,Description="Base counts in forward reads [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n'
          '##INFO=BCr,Description="Base counts in reverse reads [A:C:T:G:I:D:N:O], where D means deletion, I insertion and O other type of character">\n')
ALLELES = "ACTG"


def make_fasta(rng, contigs):
    seqs = {}
    for name, length in contigs:
        s = rng.choice(list("ACGT"), size=length)
        for _ in range(length // 40):              # homopolymer runs of 3-7
            p = int(rng.integers(0, length - 8)); s[p:p + int(rng.integers(3, 8))] = rng.choice(list("ACGT"))
        seqs[name] = "".join(s)
    return seqs


def site_row(rng, ref, deep):
    """one cell type's INFO string with internally consistent counters"""
    dp = int(np.exp(rng.uniform(np.log(5), np.log(120000 if deep else 3000))))
    nc = max(5, min(dp, int(dp * rng.uniform(0.2, 1.0)), 4000))
    scen = rng.choice(["none", "low", "het", "two", "hom", "indel", "lowsig"], p=[0.25, 0.25, 0.15, 0.1, 0.05, 0.1, 0.1])
    bc = [0] * 8
    alts = [a for a in range(4) if ALLELES[a] != ref]
    rng.shuffle(alts)
    if scen == "low":
        bc[alts[0]] = int(rng.integers(1, 4))
    elif scen == "lowsig":
        bc[alts[0]] = max(1, int(dp * rng.uniform(0.005, 0.03)))
    elif scen == "het":
        bc[alts[0]] = max(1, int(dp * rng.uniform(0.15, 0.6)))
    elif scen == "two":
        bc[alts[0]] = max(1, int(dp * rng.uniform(0.05, 0.4))); bc[alts[1]] = max(1, int(dp * rng.uniform(0.001, 0.3)))
    elif scen == "hom":
        bc[alts[0]] = max(1, int(dp * rng.uniform(0.9, 1.0)))
    elif scen == "indel":
        bc[4] = int(rng.integers(0, 3)); bc[5] = int(rng.integers(0, 4)); bc[alts[0]] = int(rng.integers(0, 2))
    bc[6] = int(rng.integers(0, 2)) if dp > 20 else 0
    bc[7] = int(rng.integers(0, 3)) if dp > 20 else 0
    over = sum(bc) - dp
    if over > 0:                                   # keep the alt classes, trim the largest
        i = int(np.argmax(bc)); bc[i] -= over
    ri = ALLELES.index(ref)
    bc[ri] += dp - sum(bc)
    cc = [0 if b == 0 else max(1, min(b, nc, int(b * rng.uniform(0.4, 1.0)) or 1)) for b in bc]
    nc = max(nc, max(cc))
    bq = [b * int(rng.integers(25, 45)) for b in bc]
    bcf = [int(rng.binomial(b, 0.5)) if b else 0 for b in bc]
    bcr = [b - f for b, f in zip(bc, bcf)]
    j = lambda v: ":".join(str(x) for x in v[:6])
    return "|".join([str(dp), str(nc), j(cc), j(bc), j(bq), j(bcf), j(bcr)])


def make_counts(rng, seqs):
    rows = {"Cancer": [], "Non-Cancer": []}
    for chrom in sorted(seqs):                     # python string order, like the reference's outputs
        length = len(seqs[chrom])
        pos = sorted(set(int(p) for p in rng.integers(1, length + 1, size=170)) | ({2, 3, 5, 6, 7, length - 2, length} if chrom == "chr1" else set()))
        for p in pos:
            ref = seqs[chrom][p - 1]
            present = [rng.random() < 0.85, rng.random() < 0.85]
            if not any(present):
                present[int(rng.integers(0, 2))] = True
            for ct, pr in zip(("Cancer", "Non-Cancer"), present):
                if pr:
                    rows[ct].append("\t".join([chrom, str(p), ref, "DP|NC|CC|BC|BQ|BCf|BCr", site_row(rng, ref, chrom == "chrM")]))
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(OUT, "_work")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    rng = np.random.default_rng(20250829)
    seqs = make_fasta(rng, [("chr1", 3000), ("chr10", 1500), ("chr2", 2000), ("chrM", 1600)])
    fasta = os.path.join(OUT, "calling.ref.fa")
    with open(fasta, "w") as f:
        for k, v in seqs.items():
            f.write(">%s\n" % k)
            for i in range(0, len(v), 60):
                f.write(v[i:i + 60] + "\n")
    rows = make_counts(rng, seqs)
    ins = []
    for ct in ("Cancer", "Non-Cancer"):
        p = os.path.join(OUT, "counts.sample.%s.tsv" % ct)
        with open(p, "w") as f:
            f.write(HEADER + "\t".join(["#CHROM", "POS", "REF", "INFO", "sample.%s" % ct]) + "\n" + "\n".join(rows[ct]) + "\n")
        ins.append(p)

    # position sets + gnomAD stand-in: drawn from the sites so that every tag occurs
    all_sites = sorted({(r.split("\t")[0], int(r.split("\t")[1])) for ct in rows for r in rows[ct]})
    pick = lambda frac: [all_sites[i] for i in sorted(rng.choice(len(all_sites), size=int(len(all_sites) * frac), replace=False))]
    sets = {"editing": pick(0.06), "pon_SR": pick(0.08), "pon_LR": pick(0.04)}
    for name, sites in sets.items():
        with open(os.path.join(OUT, "calling.%s.tsv" % name), "w") as f:
            f.write("#chrom\tpos\tinfo\n")
            for c, p in sites:
                f.write("%s\t%d\tx\n" % (c, p))
            for c in ("chr1", "chr2"):             # decoys off the sites
                f.write("%s\t%d\tx\n" % (c, 999999))
    gnomad = {}
    for c, p in pick(0.1):
        ref = seqs[c][p - 1]
        for alt in "ACGT":
            if alt != ref:
                gnomad["%s:%d:%s:%s" % (c, p, ref, alt)] = float(rng.choice([0.5, 0.02, 0.01, 0.009, 0.0001]))
    json.dump(gnomad, open(os.path.join(OUT, "calling.gnomad_af.json"), "w"), indent=0, sort_keys=True)

    install_stubs(gnomad)
    merge = load("SNVCalling/MergeBaseCellCounts.py", "ref_merge")
    step1 = load("SNVCalling/BaseCellCalling.step1.py", "ref_step1")
    step2 = load("SNVCalling/BaseCellCalling.step2.py", "ref_step2")
    step3 = load("SNVCalling/BaseCellCalling.step3.py", "ref_step3")

    # G3: merge with the explicit column order Cancer, Non-Cancer
    merged = os.path.join(OUT, "merged.tsv")
    merge.merge_cell_types_files(ins, merged)
    # G1: step1 with LongSom's config values (config/config.yaml:77-90) and script defaults
    a1, b1, a2, b2 = 0.21356677091082193, 104.95163748636298, 0.2474528917555431, 162.03696139428595
    s1 = os.path.join(OUT, "sample.calling.step1.tsv")
    step1.variant_calling_step1(merged, a1, b1, a2, b2, 2, 3, 5, 5, 2, 1, 1, 20000, s1, fasta)
    # G4: step2 (min_distance 0 as LongSom; and a second run with distance 150 to exercise "Clustered"), step3
    s2 = os.path.join(OUT, "sample.calling.step2.tsv")
    shutil.copy(s1, os.path.join(work, "in.step1.tsv"))
    step2.variant_calling_step2(os.path.join(work, "in.step1.tsv"), 0, os.path.join(OUT, "calling.editing.tsv"),
                                os.path.join(OUT, "calling.pon_SR.tsv"), os.path.join(OUT, "calling.pon_LR.tsv"), "unused", 0.01, 20000, s2)
    s2d = os.path.join(OUT, "sample.dist150.calling.step2.tsv")
    step2.variant_calling_step2(os.path.join(work, "in.step1.tsv"), 150, os.path.join(OUT, "calling.editing.tsv"),
                                os.path.join(OUT, "calling.pon_SR.tsv"), "", "unused", 0.01, 20000, s2d)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        step3.variant_calling_step3(s2, os.path.join(OUT, "sample"), 0.05, 0.3, "True", 3, 2, 10000)
        step3.variant_calling_step3(s2d, os.path.join(OUT, "sample.dist150"), 0.05, 0.3, "True", 3, 2, 150)

    # G2: the beta-binomial table, straight from scipy (the third-party arithmetic of step1.py:196,201,329)
    from scipy.stats import betabinom
    tab = []
    for (a, b) in ((a1, b1), (a2, b2)):
        for n in (5, 6, 9, 17, 32, 40, 100, 257, 1000, 4000, 20000, 100000):
            ks = sorted(set([1, 2, 3, 4, 5, n // 7 + 1, n // 3 + 1, n // 2, n // 2 + 1, n - 1, n]))
            for k in ks:
                if 1 <= k <= n:
                    tab.append([a, b, n, k, str(round(betabinom.sf(k - 0.1, n, a, b), 4)), str(round(1 - betabinom.cdf(k - 0.1, n, a, b), 4))])
    json.dump(tab, open(os.path.join(OUT, "betabinom_table.json"), "w"))
    shutil.rmtree(work, ignore_errors=True)
    print("wrote", sorted(os.listdir(OUT)))


def reanno_goldens():
    """G5-G7: the re-annotation pass, by running the reference's own pandas code.
      G5  HighConfidenceCancerVariants.HCCV_SNV on sample.calling.step2.tsv, LongSom's config values and a loose variant
      G6  CellTypeReannotation (collect_* / write_*) on a synthetic per-cell genotype table + barcodes + fusion table
      G7  scipy's betabinom.sf(k - 0.001, n, alpha2, beta2), the per-cell test of HCCVSingleCellGenotype.py:204"""
    import contextlib, io
    import pandas as pd
    hccv = load("CellTypeReannotation/HighConfidenceCancerVariants.py", "ref_hccv")
    rean = load("CellTypeReannotation/CellTypeReannotation.py", "ref_reanno")
    s2 = os.path.join(OUT, "sample.calling.step2.tsv")
    for tag, args in (("sample", (50, 0.2, 0.25, 10000)), ("sample.loose", (5, 0.05, 0.1, 400))):
        out = os.path.join(OUT, tag + ".HCCV.tsv")
        for suffix in ("", "2", "3"):
            if os.path.exists(out + suffix):
                os.remove(out + suffix)
        with contextlib.redirect_stdout(io.StringIO()):
            hccv.HCCV_SNV(s2, out, *args)
    # G6: a genotype table over 12 sites x 60 cells
    rng = np.random.default_rng(77)
    cells = ["".join(rng.choice(list("ACGT"), size=16)) for _ in range(60)]
    bar = os.path.join(OUT, "reanno.barcodes.tsv")
    with open(bar, "w") as f:
        f.write("Index\tCell_type\tNote\n")
        for i, c in enumerate(cells):
            f.write("%s\t%s\tn%d\n" % (c, "Cancer" if rng.random() < 0.4 else "Non-Cancer", i))
    geno = os.path.join(OUT, "reanno.SNVs.SingleCellGenotype.tsv")
    with open(geno, "w") as f:
        f.write("\t".join(["#CHROM", "Start", "End", "REF", "ALT_expected", "Cell_type_expected", "Num_cells_expected", "CB",
                           "Cell_type_observed", "Dp", "ALT", "VAF", "BetaBin", "MutationStatus"]) + "\n")
        for site in range(12):
            chrom = "chrM" if site >= 10 else "chr%d" % (1 + site % 3)
            pos = 1000 + 37 * site
            for ci, c in enumerate(cells):
                cov = rng.random() < (0.15 + 0.7 * (ci % 5) / 4)
                if not cov:
                    row = ["0", "0", ".", ".", "NoCoverage"]
                else:
                    dp = int(rng.integers(1, 30)); alt = int(rng.binomial(dp, 0.5)) if rng.random() < 0.45 else 0
                    if alt == 0:
                        row = [str(dp), "0", "0.0", ".", "NoAltReads"]
                    else:
                        st = rng.choice(["PASS", "BetaBin_problem", "LowVAFChrM"], p=[0.7, 0.2, 0.1])
                        row = [str(dp), str(alt), str(round(alt / dp, 4)), "." if st == "LowVAFChrM" else str(round(float(rng.random()) * 0.02, 4)), str(st)]
                f.write("\t".join([chrom, str(pos), str(pos), "A", "G", "Cancer", "7", c, "Cancer"] + row) + "\n")
    fus = os.path.join(OUT, "reanno.Fusions.SingleCellGenotype.tsv")
    with open(fus, "w") as f:
        f.write("#FusionName\tBC\tLeftBreakpoint\n")
        for i in (3, 3, 8, 21, 21, 40, 59):
            f.write("GENEA--GENEB\t%s\tchr1:%d\n" % (cells[i], i))
        f.write("GENEC--GENED\t%s\tchr2:5\n" % cells[3])
        f.write("GENEC--GENED\tTTTTTTTTTTTTTTTT\tchr2:5\n")
    for tag, fusion_file, mv, mf in (("reanno", fus, 3, 0.25), ("reanno.nofusion", "", 2, 0.5)):
        mutated, cov, cov_min = rean.collect_cells_with_SNVs(geno, mv)
        with_f = rean.collect_cells_with_fusions(fusion_file) if fusion_file else []
        cancer = rean.collect_cancer_cells(mutated, with_f, cov, mv, mf)
        rean.write_reannotated_cell_types(cancer, cov_min, bar, os.path.join(OUT, tag + ".ReannotatedCellTypes.tsv"))
    # G7
    from scipy.stats import betabinom
    a2, b2 = 0.260288007167716, 173.94711910763732
    tab = []
    for n in (1, 2, 3, 5, 8, 13, 30, 64, 65, 200, 1500, 40000):
        for k in sorted(set([1, 2, 3, n // 4 + 1, n // 2, n // 2 + 1, n - 1, n])):
            if 1 <= k <= n:
                tab.append([n, k, str(round(betabinom.sf(k - 0.001, n, a2, b2), 4))])
    json.dump({"alpha2": a2, "beta2": b2, "rows": tab}, open(os.path.join(OUT, "betabinom_sf_table.json"), "w"))
    print("wrote the re-annotation goldens")


if __name__ == "__main__":
    if "--reanno-only" in sys.argv:
        reanno_goldens()
    else:
        main()
        reanno_goldens()
