#!/usr/bin/env python3
"""Pileup-stage goldens: RUN the reference's own SplitBamCellTypes.py and BaseCellCounter.py (unmodified, imported from
/root/reference, no bytecode written) on small BAMs and commit what they write.  Runs only in the build container.

pysam / pybedtools are not installed: tools/minipysam.py provides column-replay stand-ins (its header says exactly which
third-party semantics it restates: SURVEY §8a rows a4-a5 — the CIGAR -> column step — stay hand-derived; everything the
reference itself does with a column is the reference's code: EasyReadPileup, the counting loop and gates, the row text,
the window temp files and their concatenation order, meta_to_dict, split_bam's routing and report).

Writes under tests/golden/:
  pileup.kat.json              per known-answer case (tests/kat_pileup_cases.py): the two per-cell-type tables and the
                               SplitBam report exactly as the reference wrote them (null = the reference wrote no file)
  pileup.kat_legacy.json       the cases of LEGACY_CASES through the stand-in's htslib <= 1.10 mode (minipysam.LEGACY_DEL_MERGE)
  pileup.rand.bam/.fa/.barcodes.tsv   a seeded random multi-contig sample (chr1 crosses the 50 001 window edge; contigs
                               chr1, chr10, chr2, chrM for the file order; every CIGAR op, N / IUPAC bases, low qualities,
                               all flags, MAPQ 0-60, CB missing / unknown / with "-1")
  pileup.rand.<celltype>.tsv, pileup.rand.report.txt      what the reference chain wrote for it
  pileup.randsfx.*             the same reads with "-1"-suffixed CB tags and barcodes.tsv entries
  pileup.cap.*                 a small deep pile counted with max_depth = 8 (reference's pileup call patched ONLY in that
                               one keyword through the stand-in, see run_counter(max_depth=...))
  (--check-with-real-pysam, below, writes nothing there)
  pileup.rand.HCCV.tsv, pileup.rand.genotype.{All,Alt}.tsv, pileup.randsfx.genotype.All.tsv
                               per-cell genotyping: the reference's HCCVSingleCellGenotype.py run on the UNSPLIT
                               random BAM at target sites drawn from its own count tables (every printed class as the expected
                               alt, chrM sites, sites either side of the 50 kb bin edge)

--check-with-real-pysam: ONE command away from a real pin of SURVEY §8a rows a4-a5.  Where `import pysam` and `import pybedtools`
succeed (they do not in the build container: exit status 3 and a message), the SAME driver runs over the real libraries instead of
tools/minipysam.py, writes into a scratch directory and compares every file with the committed one under tests/golden/: exit status 0
when all agree, 1 with the list of those that differ.  The fixtures whose pile is capped (pileup.cap*, made by passing another
max_depth to the stand-in's pileup call) are left out: a Cython method cannot be patched without touching the reference's source.
"""
import contextlib
import importlib.util
import io
import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/workflow/scripts"
OUT = os.path.join(ROOT, "tests", "golden")

import minipysam  # noqa: E402
from tests.support import bamwrite  # noqa: E402

REAL_PYSAM = False          # --check-with-real-pysam: the reference's imports resolve to the installed pysam / pybedtools
from tests import kat_pileup_cases as K  # noqa: E402


def load(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write_fasta(path, seqs):
    with open(path, "w") as f:
        for k, v in seqs.items():
            f.write(">%s\n" % k)
            for i in range(0, len(v), 60):
                f.write(v[i:i + 60] + "\n")


def run_chain(split, counter, bam, barcodes_tsv, fasta, sample, work, min_mq=60, extra=(), max_depth=None):
    """SplitBamCellTypes.split_bam then BaseCellCounter.main per cell type, as rules/SNVCalling.smk runs them.
    Returns ({celltype: table text or None}, report text)."""
    sdir = os.path.join(work, "SplitBam"); os.makedirs(sdir)
    with contextlib.redirect_stdout(io.StringIO()):
        split.split_bam(bam, barcodes_tsv, sdir, sample, None, None, None, min_mq, 0)
    report = open(os.path.join(sdir, sample + ".report.txt")).read()
    tables = {}
    cts = sorted(f[len(sample) + 1:-4] for f in os.listdir(sdir) if f.endswith(".bam"))
    for ct in cts:
        odir = os.path.join(work, "BaseCellCounter", ct); os.makedirs(odir)
        tmp = os.path.join(odir, "temp")
        argv = ["BaseCellCounter.py", "--bam", os.path.join(sdir, "%s.%s.bam" % (sample, ct)), "--ref", fasta, "--chrom", "all",
                "--out_folder", odir, "--nprocs", "1", "--min_mq", str(min_mq), "--tmp_dir", tmp] + list(extra)
        old_argv, old_pileup = sys.argv, minipysam.AlignmentFile.pileup
        if max_depth is not None and REAL_PYSAM:   # (a Cython method cannot be patched: the capped tables are not made over the real pysam)
            tables[ct] = None
            continue
        if max_depth is not None:          # the reference hard-codes max_depth = 200000 (BaseCellCounter.py:191): the cap fixture lowers it
            def capped(self, *a, **k):
                k["max_depth"] = max_depth
                return old_pileup(self, *a, **k)
            minipysam.AlignmentFile.pileup = capped
        try:
            sys.argv = argv
            with contextlib.redirect_stdout(io.StringIO()):
                counter.main()
        finally:
            sys.argv = old_argv
            minipysam.AlignmentFile.pileup = old_pileup
        p = os.path.join(odir, "%s.%s.tsv" % (sample, ct))
        tables[ct] = open(p).read() if os.path.exists(p) else None
    return tables, report


def strip_report(text):
    """drop the wall-clock Total_time column"""
    head, row = text.rstrip("\n").split("\n")
    h, r = head.split("\t"), row.split("\t")
    keep = [i for i, n in enumerate(h) if n != "Total_time"]
    return "\t".join(h[i] for i in keep) + "\n" + "\t".join(r[i] for i in keep) + "\n"


def strip_date(text):
    return None if text is None else "".join(l for l in text.splitlines(True) if not l.startswith("##fileDate="))


# ---- random sample ----------------------------------------------------------------------------------
def random_reads(rng, contigs, seqs, barcodes, n_clusters=9, per_cluster=34):
    reads = []
    bases = "ACGT"
    for tid, (name, length) in enumerate(contigs):
        centres = [int(c) for c in rng.integers(200, max(201, length - 400), size=n_clusters)]
        if name == "chr1":
            centres += [0, 49_960, 49_990, length - 160]
        if name == "chrM":
            centres += [0, length - 120]
        for c in centres:
            for _ in range(per_cluster):
                pos = max(0, c + int(rng.integers(-30, 40)))
                # CIGAR: leading clips, then M/=/X blocks separated by I / D / N / P, trailing clips
                ops, ref_len, q_len = [], 0, 0
                if rng.random() < 0.15: ops.append(("H", int(rng.integers(1, 5))))
                if rng.random() < 0.25: l = int(rng.integers(1, 7)); ops.append(("S", l)); q_len += l
                n_blocks = int(rng.integers(1, 5))
                for b in range(n_blocks):
                    l = int(rng.integers(1, 45)); op = "M" if rng.random() < 0.8 else ("=" if rng.random() < 0.5 else "X")
                    ops.append((op, l)); ref_len += l; q_len += l
                    if b + 1 < n_blocks:
                        r = rng.random()
                        if r < 0.25: l = int(rng.integers(1, 4)); ops.append(("I", l)); q_len += l
                        elif r < 0.5: l = int(rng.integers(1, 5)); ops.append(("D", l)); ref_len += l
                        elif r < 0.62: l = int(rng.integers(5, 60)); ops.append(("N", l)); ref_len += l
                        elif r < 0.70: ops.append(("D", 1)); ops.append(("D", 2)); ref_len += 3
                        elif r < 0.78: ops.append(("P", 1)); l = int(rng.integers(1, 3)); ops.append(("I", l)); q_len += l
                        elif r < 0.84: ops.append(("D", 2)); ops.append(("I", 1)); ref_len += 2; q_len += 1
                        elif r < 0.90: ops.append(("I", 1)); ops.append(("D", 2)); ref_len += 2; q_len += 1
                        elif r < 0.95: ops.append(("N", int(rng.integers(5, 30)))); ops.append(("I", 1)); ref_len += ops[-2][1]; q_len += 1
                        # else: two match blocks back to back (e.g. 5M3X)
                if rng.random() < 0.25: l = int(rng.integers(1, 7)); ops.append(("S", l)); q_len += l
                if pos + ref_len > length:
                    continue
                # sequence: mostly the reference with mismatches, a few N and IUPAC letters
                seq, x = [], pos
                for op, l in ops:
                    if op in "M=X":
                        for j in range(l):
                            r = rng.random(); rb = seqs[name][x + j].upper()
                            seq.append(rb if (r < 0.9 and rb in bases) else (bases[int(rng.integers(0, 4))] if r < 0.97 else ("N" if r < 0.985 else "RYKM"[int(rng.integers(0, 4))])))
                        x += l
                    elif op in "DN":
                        x += l
                    elif op in "IS":
                        seq += [bases[int(rng.integers(0, 4))] for _ in range(l)]
                qual = [int(rng.integers(20, 61)) if rng.random() < 0.88 else int(rng.integers(2, 20)) for _ in range(q_len)]
                r = rng.random()
                flag = 0x10 if rng.random() < 0.5 else 0
                if r < 0.03: flag |= 0x800
                elif r < 0.05: flag |= 0x100
                elif r < 0.065: flag |= 0x400
                elif r < 0.075: flag |= 0x200
                elif r < 0.085: flag |= 0x1
                elif r < 0.10: flag |= 0x3
                mapq = 60 if rng.random() < 0.9 else int(rng.integers(0, 60))
                r = rng.random()
                cb = None if r < 0.03 else ("ZZTOP" if r < 0.06 else barcodes[int(rng.integers(0, len(barcodes)))])
                reads.append(dict(tid=tid, pos=pos, cigar="".join("%d%s" % (l, op) for op, l in ops), seq="".join(seq), qual=qual, flag=flag, mapq=mapq,
                                  tags={} if cb is None else {"CB": cb}, name="q%d" % len(reads)))
    reads.sort(key=lambda r: (r["tid"], r["pos"]))
    return reads


def check_with_real_pysam():
    """the driver over the real pysam / pybedtools, compared with the committed fixtures; see the header"""
    global OUT, REAL_PYSAM
    try:
        import pysam            # noqa: F401
        import pybedtools       # noqa: F401
    except ImportError as e:
        print("make_pileup_goldens.py --check-with-real-pysam: %s - pysam and pybedtools (with the bedtools binary) are needed for this check; "
              "nothing was run, the committed fixtures stay pinned by tools/minipysam.py only" % e, file=sys.stderr)
        return 3
    import filecmp
    committed = OUT
    OUT = tempfile.mkdtemp(prefix="plpgold_real_")
    REAL_PYSAM = True
    try:
        for f in os.listdir(committed):                 # inputs the later stages read back from OUT (BAMs the driver wrote are rewritten identically)
            if f.startswith("pileup.") and f.endswith((".fa", ".fa.fai")):
                shutil.copy(os.path.join(committed, f), os.path.join(OUT, f))
        skipped = ["pileup.cap.Cancer.tsv, pileup.capw.Cancer.tsv (and the capped genotype table of tools/make_genotype_cap_golden.py): max_depth is hard-coded in the reference"]
        main(install_stand_in=False)
        made = sorted(f for f in os.listdir(OUT) if os.path.isfile(os.path.join(OUT, f)))
        differ = [f for f in made if not f.endswith((".bai", ".fai")) and (not os.path.exists(os.path.join(committed, f)) or not filecmp.cmp(os.path.join(OUT, f), os.path.join(committed, f), shallow=False))]
        print("compared %d files written over the real pysam with tests/golden/: %d differ%s" % (len(made), len(differ), (": " + ", ".join(differ)) if differ else ""))
        for why in skipped:
            print("not made:", why)
        return 1 if differ else 0
    finally:
        shutil.rmtree(OUT, ignore_errors=True)
        OUT = committed
        REAL_PYSAM = False


def main(install_stand_in=True):
    if install_stand_in:
        minipysam.install()
    split = load("PreProcessing/SplitBamCellTypes.py", "ref_splitbam")
    counter = load("SNVCalling/BaseCellCounter.py", "ref_counter")
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp(prefix="plpgold_")
    try:
        # ---- 1. the known-answer cases through the reference's code
        kdir = os.path.join(work, "kat"); os.makedirs(kdir)
        fa = os.path.join(kdir, "chrK.fa"); write_fasta(fa, {K.CONTIG[0]: K.REF})
        bc = os.path.join(kdir, "barcodes.tsv")
        with open(bc, "w") as f:
            f.write("Index\tCell_type\n" + "".join("%s\t%s\n" % b for b in K.BARCODES))
        kat = {}
        for name, case in sorted(K.CASES.items()):
            bam = os.path.join(kdir, name + ".bam")
            bamwrite.write_bam(bam, [K.CONTIG], sorted(case["reads"], key=lambda r: r["pos"]))
            p = case["params"]
            extra = ["--min_dp", str(p["min_dp"]), "--min_cc", str(p["min_cc"]), "--min_bq", str(p["min_bq"])]
            tables, report = run_chain(split, counter, bam, bc, fa, "kat", os.path.join(kdir, name), p["min_mq"], extra)
            kat[name] = {"params": p, "report": strip_report(report), "tables": {ct: strip_date(t) for ct, t in tables.items()}}
        json.dump(kat, open(os.path.join(OUT, "pileup.kat.json"), "w"), indent=1, sort_keys=True)
        # ... and the cases whose tables depend on the htslib version, through the stand-in's htslib <= 1.10 mode
        minipysam.LEGACY_DEL_MERGE = True
        try:
            legacy = {}
            for name, case in sorted(K.LEGACY_CASES.items()):
                bam = os.path.join(kdir, name + ".legacy.bam")
                bamwrite.write_bam(bam, [K.CONTIG], sorted(case["reads"], key=lambda r: r["pos"]))
                p = case["params"]
                extra = ["--min_dp", str(p["min_dp"]), "--min_cc", str(p["min_cc"]), "--min_bq", str(p["min_bq"])]
                tables, report = run_chain(split, counter, bam, bc, fa, "kat", os.path.join(kdir, name + ".legacy"), p["min_mq"], extra)
                legacy[name] = {"params": p, "report": strip_report(report), "tables": {ct: strip_date(t) for ct, t in tables.items()}}
            json.dump(legacy, open(os.path.join(OUT, "pileup.kat_legacy.json"), "w"), indent=1, sort_keys=True)
        finally:
            minipysam.LEGACY_DEL_MERGE = False

        # ---- 2. the random multi-contig sample (and its "-1" suffix twin)
        rng = np.random.default_rng(20251004)
        contigs = [("chr1", 60_000), ("chr10", 2_400), ("chr2", 3_000), ("chrM", 1_800)]
        seqs = {}
        for name, length in contigs:
            s = rng.choice(list("ACGT"), size=length)
            for _ in range(length // 300):
                p = int(rng.integers(0, length - 4)); s[p:p + int(rng.integers(1, 4))] = "N"
            s = "".join(s)
            seqs[name] = "".join(ch.lower() if (i // 97) % 5 == 0 else ch for i, ch in enumerate(s))      # soft-masked stretches: the reference upper-cases
        cells = ["AAAC%04dGG" % i for i in range(14)] + ["TTTG%04dCC" % i for i in range(11)]
        types = ["Cancer"] * 14 + ["Non-Cancer"] * 11
        reads = random_reads(rng, contigs, seqs, cells)
        for tag, sfx in (("rand", ""), ("randsfx", "-1")):
            rs = [dict(r, tags={k: v + sfx for k, v in r["tags"].items()}) for r in reads]
            bam = os.path.join(OUT, "pileup.%s.bam" % tag)
            bamwrite.write_bam(bam, contigs, rs)
            fa = os.path.join(OUT, "pileup.rand.fa"); write_fasta(fa, seqs)
            bc = os.path.join(OUT, "pileup.%s.barcodes.tsv" % tag)
            with open(bc, "w") as f:
                f.write("Index\tCell_type\n" + "".join("%s%s\t%s\n" % (c, sfx, t) for c, t in zip(cells, types)))
                f.write("%s%s\t%s\n" % (cells[3], sfx, "Non-Cancer"))          # a duplicated barcode: the last row wins (to_dict)
            tables, report = run_chain(split, counter, bam, bc, fa, "s", os.path.join(work, tag))
            for ct, t in tables.items():
                open(os.path.join(OUT, "pileup.%s.%s.tsv" % (tag, ct)), "w").write(strip_date(t))
            open(os.path.join(OUT, "pileup.%s.report.txt" % tag), "w").write(strip_report(report))
            print(tag, {ct: (t.count("\n") - 8 if t else None) for ct, t in tables.items()}, strip_report(report).split("\n")[1])

        # ---- 3. depth cap: 30 reads on one spot, max_depth = 8
        cap_reads = []
        for i in range(40):
            pos = 100 + (i // 4)                       # four reads start at every position 100..109
            cap_reads.append(dict(tid=0, pos=pos, cigar="30M", seq=seqs["chr2"][pos:pos + 30].upper().replace("N", "A"), qual=[30] * 30, flag=0x10 if i % 3 == 0 else 0,
                                  mapq=60, tags={"CB": cells[i % 14]}, name="c%d" % i))
        bam = os.path.join(OUT, "pileup.cap.bam")
        bamwrite.write_bam(bam, contigs, cap_reads)
        bc = os.path.join(OUT, "pileup.rand.barcodes.tsv")
        tables, report = run_chain(split, counter, bam, bc, os.path.join(OUT, "pileup.rand.fa"), "s", os.path.join(work, "cap"), max_depth=8)
        if not REAL_PYSAM:
            open(os.path.join(OUT, "pileup.cap.Cancer.tsv"), "w").write(strip_date(tables["Cancer"]))
        tables_u, _ = run_chain(split, counter, bam, bc, os.path.join(OUT, "pileup.rand.fa"), "s", os.path.join(work, "cap_u"))
        open(os.path.join(OUT, "pileup.capoff.Cancer.tsv"), "w").write(strip_date(tables_u["Cancer"]))
        print("cap rows", None if REAL_PYSAM else tables["Cancer"].count("\n") - 9, "uncapped", tables_u["Cancer"].count("\n") - 9)
        # ---- 3b. a capped pile that STRADDLES the 50 001 window edge (max_depth = 8): the reference opens a pileup per 50 kb window
        # (BaseCellCounter.py:185-191), so the columns below 50 001 are counted out of window 1's buffer - long and short reads that
        # start just before the edge - and the columns from 50 001 on out of window 2's, which only ever holds the long ones: reads the
        # first pileup drops are counted by the second
        capw_reads = []
        for i in range(48):
            pos = 49_960 + (i // 6) * 4                   # six reads at each of 49960, 49964, ... 49988 (0-based)
            ln = 110 if i % 3 == 0 else 12                # every third one reaches over the edge, the others end before it
            capw_reads.append(dict(tid=0, pos=pos, cigar="%dM" % ln, seq=seqs["chr1"][pos:pos + ln].upper().replace("N", "A"), qual=[30] * ln, flag=0x10 if i % 4 == 0 else 0,
                                   mapq=60, tags={"CB": cells[i % 14]}, name="w%d" % i))
        for i in range(12):
            pos = 50_002 + i
            capw_reads.append(dict(tid=0, pos=pos, cigar="40M", seq=seqs["chr1"][pos:pos + 40].upper().replace("N", "A"), qual=[30] * 40, flag=0, mapq=60,
                                   tags={"CB": cells[(i * 5) % 14]}, name="x%d" % i))
        capw_reads.sort(key=lambda r: (r["tid"], r["pos"]))
        bam = os.path.join(OUT, "pileup.capw.bam")
        bamwrite.write_bam(bam, contigs, capw_reads)
        tables, report = run_chain(split, counter, bam, bc, os.path.join(OUT, "pileup.rand.fa"), "s", os.path.join(work, "capw"), max_depth=8)
        if not REAL_PYSAM:
            open(os.path.join(OUT, "pileup.capw.Cancer.tsv"), "w").write(strip_date(tables["Cancer"]))
        tables_u, _ = run_chain(split, counter, bam, bc, os.path.join(OUT, "pileup.rand.fa"), "s", os.path.join(work, "capw_u"))
        open(os.path.join(OUT, "pileup.capwoff.Cancer.tsv"), "w").write(strip_date(tables_u["Cancer"]))
        print("capw rows", None if REAL_PYSAM else tables["Cancer"].count("\n") - 9, "uncapped", tables_u["Cancer"].count("\n") - 9)
        # ---- 4. per-cell genotyping at target sites: HCCVSingleCellGenotype.py on the unsplit BAMs
        geno = load("CellTypeReannotation/HCCVSingleCellGenotype.py", "ref_genotype")
        rng = np.random.default_rng(7)
        rows = [l.split("\t") for l in open(os.path.join(OUT, "pileup.rand.Cancer.tsv")).read().split("\n") if l and not l.startswith("#")]
        pick = sorted(set(rng.choice(len(rows), size=70, replace=False).tolist()) |
                      {i for i, r in enumerate(rows) if r[0] == "chr1" and int(r[1]) in (49999, 50000, 50001)} |
                      {i for i, r in enumerate(rows) if r[0] == "chrM" and int(r[1]) % 7 == 0})
        hccv = os.path.join(OUT, "pileup.rand.HCCV.tsv")
        cols = ["#CHROM", "Start", "End", "REF", "ALT", "FILTER", "Cell_types", "Up_context", "Down_context", "N_ALT", "Dp", "Nc", "Bc", "Cc", "VAF", "MCF"]
        with open(hccv, "w") as f:
            f.write("##INFO=HCCV_FILTER,Description=targets for the genotyping fixture\n" + "\t".join(cols) + "\n")
            for n, i in enumerate(pick):
                r = rows[i]
                bc = [int(x) for x in r[4].split("|")[3].split(":")]                     # BC of the six printed classes
                order = [j for j in np.argsort(bc)[::-1].tolist() if "ACTGID"[j] != r[2]]
                alt = "ACTGID"[order[0]] if bc[order[0]] > 0 and n % 3 else "ACTGIDN"[n % 7]
                if alt == r[2]:
                    alt = "N"
                f.write("\t".join([r[0], r[1], r[1], r[2], alt + (",X" if n % 5 == 0 else ""), "PASS", "Cancer", ".", ".", "1", "9", "9", "3", str(2 + n % 4), "0.3", "0.3"]) + "\n")
        for tag, flag in (("rand", "All"), ("rand", "Alt"), ("randsfx", "All")):
            out = os.path.join(OUT, "pileup.%s.genotype.%s.tsv" % (tag, flag))
            tmp = os.path.join(work, "geno_%s_%s" % (tag, flag))
            old = sys.argv
            sys.argv = ["HCCVSingleCellGenotype.py", "--bam", os.path.join(OUT, "pileup.%s.bam" % tag), "--infile", hccv, "--ref", os.path.join(OUT, "pileup.rand.fa"),
                        "--meta", os.path.join(OUT, "pileup.%s.barcodes.tsv" % tag), "--outfile", out, "--alt_flag", flag, "--nprocs", "1", "--min_mq", "60",
                        "--tmp_dir", tmp, "--chrM_contaminant", "True"]
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    geno.main()
            finally:
                sys.argv = old
            text = open(out).read()
            print("genotype", tag, flag, text.count("\n") - 1, "rows,", sum(1 for l in text.split("\n") if l.endswith("PASS")), "PASS,",
                  sum(1 for l in text.split("\n")[1:] if l and l.split("\t")[9] != "0"), "covered")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    if "--check-with-real-pysam" in sys.argv[1:]:
        sys.exit(check_with_real_pysam())
    main()
