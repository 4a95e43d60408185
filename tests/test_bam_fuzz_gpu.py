"""GPU: seeded random BAMs (every CIGAR operation incl. =, X, P, H, back-to-back D/I/N, IUPAC bases, all SAM flags, CB missing /
unknown, reads against the contig ends) through the device ingest, the host decoder and the BAM-level column oracle
(oracle/plp_oracle.c, which walks the file the way bam_plp does and shares no code with either decoder): same arrays from both
decoders, and count tables equal to the oracle's bit for bit.  The generator is the one that drew the reference-run fixture
pileup.rand.bam (tools/make_pileup_goldens.random_reads); here with other seeds, contig shapes and depths."""
import importlib.util
import os

import numpy as np
import pytest

from tests.support import bamwrite
from oracle import loader
from tests.test_ingest_gpu import both_ways

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("make_pileup_goldens", os.path.join(ROOT, "tools", "make_pileup_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # (defines functions only: the reference is touched by its main(), never here)
    return mod.random_reads


@pytest.mark.parametrize("seed", range(8))
def test_random_bam_three_ways(engine, tmp_path, seed):
    rng = np.random.default_rng(4200 + seed)
    shapes = [[("chr1", 60_000), ("chrM", 1_800)], [("chr1", 50_300), ("chr7", 700), ("chrM", 900)], [("chr1", 52_000), ("chr3", 4_099), ("chrX", 640), ("chrM", 1_500)]]
    contigs = shapes[seed % 3]
    seqs = {}
    for name, length in contigs:
        s = rng.choice(list("ACGT"), size=length)
        for _ in range(length // 300):
            p = int(rng.integers(0, length - 4)); s[p:p + int(rng.integers(1, 4))] = "N"
        seqs[name] = "".join(ch.lower() if (i // 97) % 5 == 0 else ch for i, ch in enumerate("".join(s)))
    n_cells = int(rng.choice([12, 25, 60]))
    cells = ["ACGT%05dTT" % i for i in range(n_cells)]
    celltype_of = (rng.random(n_cells) < 0.5).astype(np.uint8)
    celltype_of[0] = 0
    sfx = "-1" if seed % 2 else ""
    reads = _generator()(rng, contigs, seqs, cells, n_clusters=int(rng.choice([3, 9, 20])), per_cluster=int(rng.choice([34, 90, 150])))
    reads = [dict(r, tags={k: v + sfx for k, v in r["tags"].items()}) for r in reads]
    bam = str(tmp_path / "f.bam")
    bamwrite.write_bam(bam, contigs, reads)
    barcodes = cells                                      # (the table's barcodes are cleaned of the suffix, as read_barcodes does: SplitBamCellTypes.py:20,83)
    refs = [np.frombuffer(seqs[n].upper().encode(), dtype=np.uint8) for n, _ in contigs]          # (read_fasta upper-cases, as the reference does)
    min_mapq = int(rng.choice([60, 60, 30, 0]))
    info, dec = both_ways(engine, bam, barcodes, celltype_of, refs, min_mapq=min_mapq)
    assert info["n_records"] == len(reads)
    lens = [l for _, l in contigs]
    n_rows = 0
    for ct in range(2):
        k, r, c = engine.fetch_counts(ct)                                                       # the rows both_ways has just compared between the decoders
        ok, orf, oc = loader.plp_count(bam, barcodes, celltype_of, ct, lens, refs, min_mq=60)
        np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(r, orf); np.testing.assert_array_equal(c, oc)
        n_rows += len(k)
    assert n_rows > 1000
