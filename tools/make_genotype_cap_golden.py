#!/usr/bin/env python3
"""Genotyping golden for the pileup buffer of the UNSPLIT BAM: RUN the reference's HCCVSingleCellGenotype.py (unmodified, imported from
/root/reference, no bytecode written; pysam through tools/minipysam.py as in tools/make_pileup_goldens.py) on a small BAM whose pile is
mostly reads WITHOUT a listed barcode, with the pileup's max_depth lowered to 8 in that one keyword (the reference hard-codes 200000,
HCCVSingleCellGenotype.py:122).  The unlisted reads are never counted, but they fill the buffer: the listed reads that start behind
them at the same positions are dropped by htslib's rule - a decoder that throws unlisted reads away counts those.

Writes under tests/golden/ (contigs, reference and barcodes are pileup.rand.*'s):
  pileup.capu.bam                 30 reads without CB / with an unknown CB and 27 listed reads on chr1:3000-3080
  pileup.capu.HCCV.tsv            target sites
  pileup.capu.genotype.All.tsv    what the reference wrote with max_depth = 8
  pileup.capuoff.genotype.All.tsv ... and with its own 200000 (nothing dropped)
Runs only in the build container."""
import contextlib
import io
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(ROOT, "tests", "golden")

import minipysam  # noqa: E402
import make_pileup_goldens as M  # noqa: E402
from longsom_amd import tsvio  # noqa: E402
from tests.support import bamwrite  # noqa: E402


def main():
    minipysam.install()
    geno = M.load("CellTypeReannotation/HCCVSingleCellGenotype.py", "ref_genotype")
    names, seqs = tsvio.read_fasta(os.path.join(OUT, "pileup.rand.fa"))
    seq = dict(zip(names, (bytes(bytearray(s)).decode() for s in seqs)))
    contigs = [(n, len(seq[n])) for n in names]
    cells = [l.split("\t")[0] for l in open(os.path.join(OUT, "pileup.rand.barcodes.tsv")).read().split("\n")[1:] if l]
    chrom = "chr1"
    tid = names.index(chrom)
    ref = seq[chrom].upper().replace("N", "A")
    reads = []
    comp = {"A": "C", "C": "G", "G": "T", "T": "A"}
    for j, pos in enumerate(range(3000, 3010, 2)):          # five start positions; at each: six unlisted reads, then three listed ones
        for i in range(6):
            tags = {} if i % 2 else {"CB": "ZZZZ%04dZZ" % (j * 6 + i)}
            reads.append(dict(tid=tid, pos=pos, cigar="70M", seq=ref[pos:pos + 70], qual=[35] * 70, flag=0x10 if i % 3 == 0 else 0, mapq=60, tags=tags, name="u%d_%d" % (j, i)))
        for i in range(3):
            s = list(ref[pos:pos + 70])
            for p in (3012, 3020, 3033):                     # the listed reads carry alts at three of the target sites
                if (i + j) % 2 == 0:
                    s[p - pos] = comp[s[p - pos]]
            reads.append(dict(tid=tid, pos=pos, cigar="70M", seq="".join(s), qual=[35] * 70, flag=0, mapq=60, tags={"CB": cells[(j * 3 + i) % len(cells)]}, name="l%d_%d" % (j, i)))
    for i in range(12):                                      # listed reads that are the FIRST of their start position: never tested against the cap
        pos = 3011 + i
        s = list(ref[pos:pos + 50])
        if i % 3 == 0 and pos <= 3033 < pos + 50:
            s[3033 - pos] = comp[s[3033 - pos]]
        reads.append(dict(tid=tid, pos=pos, cigar="50M", seq="".join(s), qual=[35] * 50, flag=0, mapq=60, tags={"CB": cells[(i * 5 + 2) % len(cells)]}, name="f%d" % i))
    reads.sort(key=lambda r: (r["tid"], r["pos"]))           # (stable: the file order inside a start position is the order above)
    bam = os.path.join(OUT, "pileup.capu.bam")
    bamwrite.write_bam(bam, contigs, reads)
    hccv = os.path.join(OUT, "pileup.capu.HCCV.tsv")
    cols = ["#CHROM", "Start", "End", "REF", "ALT", "FILTER", "Cell_types", "Up_context", "Down_context", "N_ALT", "Dp", "Nc", "Bc", "Cc", "VAF", "MCF"]
    with open(hccv, "w") as f:
        f.write("##INFO=HCCV_FILTER,Description=targets for the genotyping buffer fixture\n" + "\t".join(cols) + "\n")
        for n, p in enumerate((3005, 3009, 3012, 3020, 3033, 3047, 3060, 3072)):
            f.write("\t".join([chrom, str(p + 1), str(p + 1), ref[p], comp[ref[p]], "PASS", "Cancer", ".", ".", "1", "9", "9", "3", str(2 + n % 4), "0.3", "0.3"]) + "\n")
    work = tempfile.mkdtemp(prefix="capugold_")
    try:
        for tag, cap in (("capu", 8), ("capuoff", None)):
            out = os.path.join(OUT, "pileup.%s.genotype.All.tsv" % tag)
            old_argv, old_pileup = sys.argv, minipysam.AlignmentFile.pileup
            if cap is not None:
                def capped(self, *a, **k):
                    k["max_depth"] = cap
                    return old_pileup(self, *a, **k)
                minipysam.AlignmentFile.pileup = capped
            sys.argv = ["HCCVSingleCellGenotype.py", "--bam", bam, "--infile", hccv, "--ref", os.path.join(OUT, "pileup.rand.fa"),
                        "--meta", os.path.join(OUT, "pileup.rand.barcodes.tsv"), "--outfile", out, "--alt_flag", "All", "--nprocs", "1", "--min_mq", "60",
                        "--tmp_dir", os.path.join(work, tag), "--chrM_contaminant", "True"]
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    geno.main()
            finally:
                sys.argv = old_argv
                minipysam.AlignmentFile.pileup = old_pileup
            text = open(out).read()
            print(tag, text.count("\n") - 1, "rows,", sum(int(l.split("\t")[9]) for l in text.split("\n")[1:] if l), "reads of depth in all")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
