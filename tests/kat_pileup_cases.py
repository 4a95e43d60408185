"""Hand-derived known-answer cases for the pileup stage.

The reference's pileup cannot run here (pysam / htslib absent, no test vectors in the reference), so
these cases are written BY HAND from the documented htslib bam_plp + pysam PileupColumn semantics
(SURVEY.md §8a rows a4-a8): every expected counter below was worked out on paper, not computed.
They pin oracle/plp_oracle.c, the decoder (hostio/bamio.cpp) and the HIP kernels alike.

Reference contig "chrK", 120 bp: base at 0-based position i is "ACGT"[i % 4], except position 50 = 'N'.
Barcodes BC0..BC5 are Cancer (cell type 0), BN0 is Non-Cancer (1).
Row notation: pos1 -> (DP, NC, CC, BC, BQ, BCf, BCr) with each vector a dict over "ACTGID" (missing = 0).
"""
CONTIG = ("chrK", 120)
REF = "".join("N" if i == 50 else "ACGT"[i % 4] for i in range(120))
BARCODES = [("BC0", "Cancer"), ("BC1", "Cancer"), ("BC2", "Cancer"), ("BC3", "Cancer"), ("BC4", "Cancer"), ("BC5", "Cancer"),
            ("BN0", "Non-Cancer")]
LOOSE = dict(min_bq=20, min_mq=60, min_dp=1, min_cc=1)
STRICT = dict(min_bq=20, min_mq=60, min_dp=5, min_cc=5)


def rd(pos, cigar, seq, qual, cb="BC0", flag=0, mapq=60, name=None):
    tags = {} if cb is None else {"CB": cb}
    return dict(tid=0, pos=pos, cigar=cigar, seq=seq, qual=list(qual), flag=flag, mapq=mapq, tags=tags, name=name or "r")


def one(base, q, rev=False, n=1, cells=1):
    """row with a single symbol class: n entries, `cells` distinct cells, quality sum q"""
    return (n, cells, {base: cells}, {base: n}, {base: q}, {} if rev else {base: n}, {base: n} if rev else {})


CASES = {
    "match_only": dict(
        reads=[rd(10, "8M", "GTACGTAC", [30] * 8)], params=LOOSE,
        cancer={11: one("G", 30), 12: one("T", 30), 13: one("A", 30), 14: one("C", 30), 15: one("G", 30), 16: one("T", 30),
                17: one("A", 30), 18: one("C", 30)}),
    "mismatch_reverse": dict(
        reads=[rd(10, "4M", "GTTC", [30, 31, 32, 33], cb="BC1", flag=16)], params=LOOSE,
        cancer={11: one("G", 30, rev=True), 12: one("T", 31, rev=True), 13: one("T", 32, rev=True), 14: one("C", 33, rev=True)}),
    # anchor base before an insertion is counted as I (its own letter is discarded), with its own quality
    "insertion": dict(
        reads=[rd(20, "3M2I3M", "ACGAATAC", [40, 41, 42, 10, 10, 43, 44, 45])], params=LOOSE,
        cancer={21: one("A", 40), 22: one("C", 41), 23: one("I", 42), 24: one("T", 43), 25: one("A", 44), 26: one("C", 45)}),
    # anchor before a deletion -> D; interior deletion columns -> O (counted in DP/NC, not printed) with the
    # quality of the first base AFTER the deletion (20 here)
    "deletion": dict(
        reads=[rd(30, "3M2D3M", "GTATAC", [30, 31, 32, 20, 34, 35])], params=LOOSE,
        cancer={31: one("G", 30), 32: one("T", 31), 33: one("D", 32), 34: (1, 1, {}, {}, {}, {}, {}), 35: (1, 1, {}, {}, {}, {}, {}),
                36: one("T", 20), 37: one("A", 34), 38: one("C", 35)}),
    # the base-quality gate applies to deletion columns through the next base: 19 < 20 removes both
    "deletion_lowq_next": dict(
        reads=[rd(30, "3M2D3M", "GTATAC", [30, 31, 32, 19, 34, 35])], params=LOOSE,
        cancer={31: one("G", 30), 32: one("T", 31), 33: one("D", 32), 37: one("A", 34), 38: one("C", 35)}),
    # reference skips are 'NA' (never counted); a column whose reference base is N is never emitted
    "intron_and_refN": dict(
        reads=[rd(40, "3M10N3M", "ACGCGT", [30] * 6, cb="BC0"), rd(48, "6M", "ACGTAC", [30] * 6, cb="BC1")], params=LOOSE,
        cancer={41: one("A", 30), 42: one("C", 30), 43: one("G", 30), 49: one("A", 30), 50: one("C", 30), 52: one("T", 30),
                53: one("A", 30), 54: one("C", 60, n=2, cells=2), 55: one("G", 30), 56: one("T", 30)}),
    "clips": dict(
        reads=[rd(60, "2H3S4M2S", "TTTACGTGG", [5, 5, 5, 30, 31, 32, 33, 5, 5])], params=LOOSE,
        cancer={61: one("A", 30), 62: one("C", 31), 63: one("G", 32), 64: one("T", 33)}),
    "low_bq": dict(
        reads=[rd(70, "3M", "GTA", [19, 20, 21])], params=LOOSE,
        cancer={72: one("T", 20), 73: one("A", 21)}),
    # secondary / duplicate / qcfail / unmapped / supplementary / paired-not-proper are dropped; proper pair kept
    "flags": dict(
        reads=[rd(80, "2M", "AC", [30, 30], cb="BC0", flag=0x100), rd(80, "2M", "AC", [30, 30], cb="BC0", flag=0x400),
               rd(80, "2M", "AC", [30, 30], cb="BC0", flag=0x200), rd(80, "2M", "AC", [30, 30], cb="BC0", flag=0x4),
               rd(80, "2M", "AC", [30, 30], cb="BC0", flag=0x800), rd(80, "2M", "AC", [30, 30], cb="BC1", flag=0x1),
               rd(80, "2M", "AC", [30, 30], cb="BC2", flag=0x3), rd(80, "2M", "AC", [30, 30], cb="BC3", flag=0x10)], params=LOOSE,
        cancer={81: (2, 2, {"A": 2}, {"A": 2}, {"A": 60}, {"A": 1}, {"A": 1}), 82: (2, 2, {"C": 2}, {"C": 2}, {"C": 60}, {"C": 1}, {"C": 1})}),
    "mapq": dict(
        reads=[rd(84, "2M", "AC", [30, 30], cb="BC0", mapq=59), rd(84, "2M", "AC", [30, 30], cb="BC1", mapq=60)], params=LOOSE,
        cancer={85: one("A", 30), 86: one("C", 30)}),
    # no CB tag -> dropped; "-1" suffix tolerated; unknown barcode dropped; other cell type goes to its own table
    "barcodes": dict(
        reads=[rd(90, "2M", "GT", [30, 30], cb=None), rd(90, "2M", "GT", [30, 30], cb="BC0-1"), rd(90, "2M", "GT", [30, 30], cb="ZZZ"),
               rd(90, "2M", "GT", [31, 31], cb="BN0")], params=LOOSE,
        cancer={91: one("G", 30), 92: one("T", 30)}, noncancer={91: one("G", 31), 92: one("T", 31)}),
    # an N base call is counted in DP and NC but has no printed class
    "n_base": dict(
        reads=[rd(96, "1M", "N", [30], cb="BC0"), rd(96, "1M", "A", [30], cb="BC1")], params=LOOSE,
        cancer={97: (2, 2, {"A": 1}, {"A": 1}, {"A": 30}, {"A": 1}, {})}),
    # 0-based position 0 (POS 1) of a contig is never visited (windows start at 1)
    "contig_start": dict(
        reads=[rd(0, "4M", "ACGT", [30] * 4)], params=LOOSE,
        cancer={2: one("C", 30), 3: one("G", 30), 4: one("T", 30)}),
    # default gates: count >= 5 and distinct cells >= 5
    "gates": dict(
        reads=[rd(100, "1M", "A", [30], cb="BC%d" % i) for i in range(5)] +
              [rd(102, "1M", "G", [30], cb="BC%d" % i) for i in (0, 0, 1, 2, 3)] +
              [rd(104, "1M", "A", [30], cb="BC%d" % i) for i in range(4)] +
              [rd(106, "1M", "G", [30], cb="BC%d" % i) for i in (0, 0, 1, 2, 3, 4)], params=STRICT,
        cancer={101: one("A", 150, n=5, cells=5), 107: one("G", 180, n=6, cells=5)}),
    "same_cell_multiple": dict(
        reads=[rd(110, "1M", "G", [30], cb="BC0") for _ in range(3)], params=LOOSE,
        cancer={111: one("G", 90, n=3, cells=1)}),
    # 2M1D1I2M: the M's last column is the D anchor; the single D column is the last of its op with an insertion
    # next -> counted as I, quality of the base at the query cursor (the inserted base, 25)
    "deletion_then_insertion": dict(
        reads=[rd(112, "2M1D1I2M", "ACTTA", [30, 31, 25, 33, 34])], params=LOOSE,
        cancer={113: one("A", 30), 114: one("D", 31), 115: one("I", 25), 116: one("T", 33), 117: one("A", 34)}),
    "eq_and_x_ops": dict(
        reads=[rd(4, "2=1X1=", "ACTT", [30] * 4)], params=LOOSE,
        cancer={5: one("A", 30), 6: one("C", 30), 7: one("T", 30), 8: one("T", 30)}),
    # IUPAC letters are 'NA': not counted at all
    "iupac": dict(
        reads=[rd(8, "2M", "RC", [30, 30])], params=LOOSE,
        cancer={10: one("C", 30)}),
    # 3M1D2D3M: consecutive deletions are ONE deletion of 3 since htslib 1.11 (the M's last column is its anchor, all three deleted
    # columns are interior: O with the quality of the next base); htslib <= 1.10: see LEGACY_CASES
    "consecutive_deletions": dict(
        reads=[rd(30, "3M1D2D3M", "GTATAC", [30, 31, 32, 20, 34, 35])], params=LOOSE,
        cancer={31: one("G", 30), 32: one("T", 31), 33: one("D", 32), 34: (1, 1, {}, {}, {}, {}, {}), 35: (1, 1, {}, {}, {}, {}, {}),
                36: (1, 1, {}, {}, {}, {}, {}), 37: one("T", 20), 38: one("A", 34), 39: one("C", 35)}),
    # a pad between the anchor and an insertion: the anchor is still an insertion anchor
    "pad_insertion": dict(
        reads=[rd(20, "3M1P1I3M", "ACGATAC", [40, 41, 42, 10, 43, 44, 45])], params=LOOSE,
        cancer={21: one("A", 40), 22: one("C", 41), 23: one("I", 42), 24: one("T", 43), 25: one("A", 44), 26: one("C", 45)}),
}

# htslib <= 1.10 (hostio.set_legacy_del_merge / plp_set_legacy_del_merge / minipysam.LEGACY_DEL_MERGE): the last column of a D
# operation that is followed by another D is flagged too — pysam prints "*-2NN" there and EasyReadPileup counts a D (quality of the
# next query base, like every deleted column)
LEGACY_CASES = {
    "consecutive_deletions": dict(
        reads=CASES["consecutive_deletions"]["reads"], params=LOOSE,
        cancer={31: one("G", 30), 32: one("T", 31), 33: one("D", 32), 34: one("D", 20), 35: (1, 1, {}, {}, {}, {}, {}),
                36: (1, 1, {}, {}, {}, {}, {}), 37: one("T", 20), 38: one("A", 34), 39: one("C", 35)}),
}

ORDER = "ACTGID"


def expected_rows(spec):
    """{pos1: tuple} -> {pos0: 42-word list} (N and O classes are not asserted: words 8,9,16,17,... stay free)"""
    out = {}
    for pos1, (dp, nc, cc, bc, bq, bcf, bcr) in spec.items():
        row = [0] * 42
        row[0], row[1] = dp, nc
        for off, vec in ((2, cc), (10, bc), (18, bq), (26, bcf), (34, bcr)):
            for k, v in vec.items():
                row[off + ORDER.index(k)] = v
        out[pos1 - 1] = row
    return out


PRINTED = [0, 1] + [o + i for o in (2, 10, 18, 26, 34) for i in range(6)]
