"""GPU: merge + step-1 call kernel against the golden step1 TSV written by the reference's own code."""
import os

import numpy as np
import pytest

from tests.util import neg_zero

from longsom_amd import tsvio
from longsom_amd._lib import CallParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def strip_date(text):
    return "\n".join(l for l in text.split("\n") if not l.startswith("##fileDate="))


def test_step1_matches_reference_golden(engine):
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    per_ct = [tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.%s.tsv" % ct), names)[:3] for ct in ("Cancer", "Non-Cancer")]
    engine.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
    n_sites, n_cand = engine.call_step1()
    calls = engine.fetch_calls()
    assert len(calls) == n_sites == len(np.unique(np.concatenate([p[0] for p in per_ct])))
    header = [l for l in open(os.path.join(G, "merged.tsv")) if l.startswith("##")]
    text = tsvio.format_step1_tsv(calls, per_ct, names, ["Cancer", "Non-Cancer"], header)
    want = neg_zero(open(os.path.join(G, "sample.calling.step1.tsv")).read())   # sign of fp noise, whole p-value tokens only
    got_lines, want_lines = strip_date(text).split("\n"), strip_date(want).split("\n")
    assert len(got_lines) == len(want_lines)
    bad = [(g, w) for g, w in zip(got_lines, want_lines) if g != w]
    assert not bad, "first mismatch:\n%s\n%s" % bad[0]
    # the fetch filter keeps exactly the rows step2's awk keeps (ALT != "." and FILTER != "."), step2.py:23
    cand = engine.fetch_calls(candidates_only=True)
    n_awk = sum(1 for l in want_lines if l and not l.startswith("#") and l.split("\t")[4] != "." and l.split("\t")[5] != ".")
    n_noisy_only = sum(1 for l in want_lines if l and not l.startswith("#") and l.split("\t")[4] == "." and l.split("\t")[5] != ".")
    assert len(cand) == n_awk + n_noisy_only and n_cand == n_awk


def test_betabinom_table(engine):
    """k_call's tail sums against scipy (the reference's third-party arithmetic) through single-site inputs."""
    import json
    tab = json.load(open(os.path.join(G, "betabinom_table.json")))
    a1, b1 = tab[0][0], tab[0][1]
    rows = [t for t in tab if t[0] == a1 and t[1] == b1 and t[2] >= 5]
    L = 64 * (len(rows) + 2)
    engine.set_contigs([L]); engine.load_reference(0, np.full(L, ord("A"), np.uint8))
    keys = np.arange(len(rows), dtype=np.int64) * 64 + 10
    counts = np.zeros((len(rows), 42), np.uint32)
    for i, (_, _, n, k, _, _) in enumerate(rows):
        counts[i, 0] = n; counts[i, 1] = 5
        counts[i, 10 + 0] = n - k; counts[i, 10 + 1] = k      # ref A, alt C
        counts[i, 2 + 0] = 1 if n > k else 0; counts[i, 2 + 1] = 1
    engine.load_counts([keys], [counts])
    engine.call_step1()
    calls = engine.fetch_calls()
    for c, (_, _, n, k, sf, _) in zip(calls, rows):
        got = repr(int(c["p_bc"][0][0]) / 10000.0)
        assert got == ("0.0" if sf == "-0.0" else sf), (n, k, got, sf)


def test_step2_matches_reference_golden(engine):
    import json
    from longsom_amd import calling
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    ed, sr, lr = (calling.read_posset_keys(os.path.join(G, "calling.%s.tsv" % k), names) for k in ("editing", "pon_SR", "pon_LR"))
    af = json.load(open(os.path.join(G, "calling.gnomad_af.json")))
    s1 = open(os.path.join(G, "sample.calling.step1.tsv")).read()
    got = calling.step2(s1, engine, names, ed, sr, lr, 0, af, 0.01)
    assert got == open(os.path.join(G, "sample.calling.step2.tsv")).read()
    got = calling.step2(s1, engine, names, ed, sr, calling.read_posset_keys("", names), 150, af, 0.01)
    assert got == open(os.path.join(G, "sample.dist150.calling.step2.tsv")).read()


@pytest.mark.parametrize("n_ct", [1, 3, 4])
def test_step1_other_cell_type_counts_match_oracle(engine, n_ct):
    """k_call_gather is compiled per cell-type count: 1, 3 and 4 cell types against the oracle (itself byte-identical to the
    reference on the two-cell-type golden)."""
    from oracle import calling_oracle
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    base = [tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.%s.tsv" % ct), names)[:3] for ct in ("Cancer", "Non-Cancer")]
    rng = np.random.default_rng(n_ct)
    per_ct, ct_names = [], ["T%d" % i for i in range(n_ct)]
    for i in range(n_ct):
        k, r, c = base[i % 2]
        keep = rng.random(len(k)) < 0.8                     # different site sets per cell type
        per_ct.append((k[keep], r[keep], c[keep]))
    engine.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
    n_sites, _ = engine.call_step1(CallParams.longsom_defaults(min_cell_types=min(2, n_ct)))
    calls = engine.fetch_calls()
    assert len(calls) == n_sites
    merged = tsvio.format_merged_tsv(per_ct, names, ct_names)
    header = [l + "\n" for l in merged.split("\n") if l.startswith("##")]
    got = tsvio.format_step1_tsv(calls, per_ct, names, ct_names, header)
    want = calling_oracle.step1(merged, dict(zip(names, [s.tobytes().decode() if hasattr(s, "tobytes") else s for s in seqs])),
                                min_cell_types=min(2, n_ct), info_lines=tsvio.STEP1_INFO_LINES)
    want = neg_zero(want)
    g, w = strip_date(got).split("\n"), strip_date(want).split("\n")
    if w and w[-1] != "" and g and g[-1] == "":
        g = g[:-1]
    bad = [(a, b) for a, b in zip(g, w) if a != b]
    assert len(g) == len(w) and not bad, "first mismatch:\n%s\n%s" % (bad[0] if bad else (len(g), len(w)))


def test_homopolymer_context_at_tile_and_contig_edges(engine):
    """The LC_Upstream / LC_Downstream flags (step1.py:95-107, :347-354) read five reference bases either side of a site.
    k_call_gather cuts them from one load per 64-position tile plus five bases either side of it: runs that straddle tile
    borders, sites in the first and last five bases of a contig, contigs shorter than a tile and of exactly a tile."""
    from oracle import calling_oracle
    rng = np.random.default_rng(77)
    lens = [197, 70, 6, 64, 129, 11, 65, 63]
    names = ["c%d" % i for i in range(len(lens))]
    seqs = []
    for L in lens:
        s, cur = [], rng.integers(0, 4)
        for _ in range(L):
            if rng.random() > 0.62:
                cur = rng.integers(0, 4)
            s.append("ACGT"[cur])
        seqs.append(np.frombuffer("".join(s).encode(), np.uint8).copy())
    engine.set_contigs(lens)
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    cls_of = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}
    per_ct = []
    for ct in range(2):
        keys, refs, rows = [], [], []
        for t, L in enumerate(lens):
            for pos in range(L):
                if rng.random() < 0.2:
                    continue
                r = np.zeros(42, np.uint32)
                rc = cls_of[int(seqs[t][pos])]
                bc = np.zeros(8, np.int64); bc[rc] = rng.integers(15, 60)
                for alt in rng.permutation([c for c in range(4) if c != rc])[:rng.integers(0, 3)]:
                    bc[alt] = rng.integers(1, 12)
                cc = np.minimum(bc, rng.integers(1, 9, 8)) * (bc > 0)
                r[0] = bc.sum(); r[1] = max(5, int(cc.max()) + 3); r[2:10] = cc; r[10:18] = bc; r[26:34] = bc
                keys.append((t << 32) | pos); refs.append(int(seqs[t][pos])); rows.append(r)
        per_ct.append((np.asarray(keys, np.int64), np.asarray(refs, np.uint8), np.stack(rows)))
    ct_names = ["Cancer", "Non-Cancer"]
    engine.load_counts([p[0] for p in per_ct], [p[2] for p in per_ct])
    n_sites, _ = engine.call_step1()
    calls = engine.fetch_calls()
    assert len(calls) == n_sites
    merged = tsvio.format_merged_tsv(per_ct, names, ct_names)
    header = [l + "\n" for l in merged.split("\n") if l.startswith("##")]
    got = tsvio.format_step1_tsv(calls, per_ct, names, ct_names, header)
    want = neg_zero(calling_oracle.step1(merged, dict(zip(names, [s.tobytes().decode() for s in seqs])), info_lines=tsvio.STEP1_INFO_LINES))
    g, w = strip_date(got).split("\n"), strip_date(want).split("\n")
    if w and w[-1] != "" and g and g[-1] == "":
        g = g[:-1]
    bad = [(a, b) for a, b in zip(g, w) if a != b]
    assert len(g) == len(w) and not bad, "first mismatch:\n%s\n%s" % (bad[0] if bad else (len(g), len(w)))
    n_up, n_down = sum("LC_Upstream" in l for l in w), sum("LC_Downstream" in l for l in w)
    assert n_up > 20 and n_down > 20, (n_up, n_down)


def test_a_table_set_again_after_installed_counts_is_set(engine):
    """lsg_load_counts changes the handle's number of cell types: the table it held before is not 'the table already set' any more, and
    setting it again takes effect (the same-table shortcut is decided inside the library, against what the handle holds)"""
    import numpy as np
    from longsom_amd import synth
    m = synth.named("C1", n_reads=4000, n_genes=60, n_cb=50)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    rows, cols = engine.pileup_count()
    k0, r0, c0 = engine.fetch_counts(0)
    engine.load_counts([k0], [c0])                                # one cell type installed: n_ct = 1 inside the handle
    assert engine.n_ct == 1
    engine.set_barcodes(m.celltype_of, 2)                         # the very table of before
    assert engine.n_ct == 2
    engine.synth_reads(m)
    assert engine.pileup_count() == (rows, cols)
    k1, r1, c1 = engine.fetch_counts(1)
    assert len(k1) == rows[1]


def test_call_and_tables_refuse_a_contig_without_its_reference(engine):
    """installed count rows (lsg_load_counts) come without the count's own check: the call reads every site's reference base and its
    context, the tables print them - a contig whose bases were never loaded is an error, not a fault on the device"""
    names, seqs = tsvio.read_fasta(os.path.join(G, "calling.ref.fa"))
    engine.set_contigs([len(s) for s in seqs])                      # (a new contig table: no references yet)
    k, r, c = tsvio.parse_counts_tsv(os.path.join(G, "counts.sample.Cancer.tsv"), names)[:3]
    engine.load_counts([k], [c])
    with pytest.raises(RuntimeError, match="reference of contig 0 not loaded"):
        engine.call_step1()
    engine.set_table_names(names, ["Cancer"])
    with pytest.raises(RuntimeError, match="reference of contig 0 not loaded"):
        engine.format_table(0)
    for t, s in enumerate(seqs):
        engine.load_reference(t, s)
    engine.load_counts([k], [c])                                     # (a new reference drops the counts made against the old one)
    assert engine.call_step1()[0] > 0 and engine.format_table(0) > 0
    engine.free_table()
