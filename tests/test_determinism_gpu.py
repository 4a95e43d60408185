"""GPU: counting the same resident reads twice gives the same rows, word for word, on a sample large enough to load the whole
chip (several million rows).  Parity against the oracle is checked on samples the CPU finishes in seconds; a timing-dependent fault
— this test exists because of one: 128-bit buffer stores whose data registers were reused too early on gfx950 corrupted ~1e-4 of the
rows under load and nothing else noticed — only shows at scale, where run-to-run identity is the property that can be checked."""
import json
import os

import numpy as np
import pytest
import xxhash

from longsom_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIN = os.path.join(G, "rows_hash_c4_2500k.json")                  # call records of this sample: written by the HIP path itself (see below)
ORACLE_C4 = os.path.join(G, "rows_hash_oracle_c4_2500000.json")   # count rows: written by the CPU oracle (tools/oracle_hashes.py), never by the GPU
ORACLE_C2 = os.path.join(G, "rows_hash_oracle_c2_1500000.json")


def digest(eng):
    out = {}
    for ct in range(2):
        k, r, c = eng.fetch_counts(ct)
        out["ct%d" % ct] = [xxhash.xxh64(np.ascontiguousarray(x).tobytes()).hexdigest() for x in (k, r, c)]
    calls = eng.fetch_calls(candidates_only=True)
    out["calls"] = [len(calls), xxhash.xxh64(calls.tobytes()).hexdigest()]
    return out

pytestmark = pytest.mark.gpu


def oracle_calls(tag):
    d = json.load(open(os.path.join(G, "calls_hash_oracle_%s.json" % tag)))
    return d["candidate_rows"], {c: tuple(v) for c, v in d["per_contig"].items()}


def candidate_text_digest(eng, m, candidates_only=False):
    """(rows, {contig: (rows, xxhash of its rows' text)}) of the step-1 text of the rows step 2 keeps, for the counts and calls resident
    in eng.  Per contig: the writer puts the contigs in Python string order (the reference's file order), the oracle tool in header order.
    candidates_only: fetch and format the candidate sites' records only (the rows step 2 keeps are among them: a row without an ALT is
    dropped by its awk filter, step2.py:23) - what the full-size workloads can afford."""
    from longsom_amd import tsvio
    per_ct = [eng.fetch_counts(ct) for ct in range(2)]
    calls = eng.fetch_calls(candidates_only=candidates_only)
    if candidates_only:                                            # (the writer wants a call record for every merged site of the rows it is given: the candidate sites' rows)
        ck = np.ascontiguousarray(calls["key"])
        sub = []
        for k, r, c in per_ct:
            i = np.searchsorted(ck, k)
            keep = (i < len(ck)) & (ck[np.minimum(i, len(ck) - 1)] == k)
            sub.append((k[keep], r[keep], c[keep]))
        per_ct = sub
    text = tsvio.write_step1_tsv("/dev/null", calls, per_ct, m.contig_names, ["Cancer", "Non-Cancer"], [], header=False, as_bytes=True)
    return text_digest(text, m)


def device_text_digest(eng, m):
    """the same digest of the same rows as the DEVICE prints them (csrc/tables.hip: table TABLE_STEP1_KEPT, what the fused chain's step 2
    starts from): its text never passes through the host writer"""
    eng.set_table_names(m.contig_names, ["Cancer", "Non-Cancer"])
    n = eng.format_table(eng.TABLE_STEP1_KEPT)
    text = eng.table_bytes(eng.TABLE_STEP1_KEPT, n)
    eng.free_table()
    return text_digest(text, m)


def text_digest(text, m):
    from longsom_amd import tsvio
    sc = tsvio.scan_rows(text, m.contig_names)
    tid = sc.key >> 32
    out = {}
    for t in np.unique(tid).tolist():
        i = np.nonzero(tid == t)[0]
        assert (np.diff(i) == 1).all(), "a contig's rows are contiguous in the table"
        a, b = int(sc.off[i[0]]), int(sc.off[i[-1]] + sc.len[i[-1]] + 1)
        out[m.contig_names[t]] = (len(i), xxhash.xxh64(text[a:b]).hexdigest())
    return sc.n_rows, out


def test_candidate_call_records_equal_the_cpu_oracles_small(engine):
    """C1 at 20 k reads (263 k merged sites, 33.6 k candidate rows): count + merge + step 1 on the GPU == the CPU oracles' text"""
    m = synth.named("C1", n_reads=20_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    engine.pileup_count(); engine.call_step1()
    assert candidate_text_digest(engine, m) == oracle_calls("c1_20000")
    assert device_text_digest(engine, m) == oracle_calls("c1_20000")


def test_two_counts_of_the_same_reads_are_identical(engine):
    m = synth.named("C4", n_reads=2_500_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    ref = None
    for it in range(3):
        rows, cols = engine.pileup_count()
        n_sites, n_cand = engine.call_step1()
        got = [engine.fetch_counts(ct) for ct in range(2)]
        if ref is None:
            ref = (rows, cols, n_sites, n_cand, got)
            assert sum(rows) > 5_000_000
            continue
        assert (rows, cols, n_sites, n_cand) == ref[:4]
        for ct in range(2):
            assert np.array_equal(got[ct][0], ref[4][ct][0])
            bad = np.nonzero((got[ct][2] != ref[4][ct][2]).any(axis=1))[0]
            assert len(bad) == 0, "cell type %d: %d rows differ between two counts, first at key %d" % (ct, len(bad), int(got[ct][0][bad[0]]))
    # ... and they are the rows the CPU ORACLE counts for this sample: tools/oracle_hashes.py evaluated the same model on the host
    # (hostio.synth_records == the device generator, tests/test_synth_gpu.py), counted it with oracle/count_oracle.c on all cores of
    # the build container and committed the hashes; nothing the GPU wrote is part of that pin
    d = digest(engine)
    want = json.load(open(ORACLE_C4))
    assert rows == want["rows"] and cols == want["columns"]
    assert d["ct0"] == want["ct0"] and d["ct1"] == want["ct1"], "count rows of the 2.5 M-read sample differ from the CPU oracle's"
    # the candidate call records of the sample: their step-1 text (the product's native writer, pinned byte for byte to the reference's
    # goldens) hashes to what the CPU oracles wrote for this sample — count_oracle.c + calling_oracle.py step 1 (scipy) in 6 processes,
    # tools/oracle_call_hash.py; nothing the GPU wrote is part of the pin
    want_calls = oracle_calls("c4_2500000")
    assert candidate_text_digest(engine, m) == want_calls
    assert candidate_text_digest(engine, m, candidates_only=True) == want_calls          # (the form the full-size test uses)
    assert device_text_digest(engine, m) == want_calls                                   # (and as the device prints the rows)
    # (the digest of the raw call records the HIP path itself wrote in round 1: kept as a cross-build check)
    if os.environ.get("LSG_WRITE_PIN") == "1":
        json.dump({"calls": d["calls"]}, open(PIN, "w"), indent=1)
    assert d["calls"] == json.load(open(PIN))["calls"]


def test_c2_sample_rows_equal_the_cpu_oracle(engine):
    """BASELINE's C2 workload at 1.5 M reads (16.9 M rows): the HIP rows hash to what the region-parallel CPU oracle wrote."""
    m = synth.named("C2", n_reads=1_500_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    rows, cols = engine.pileup_count()
    want = json.load(open(ORACLE_C2))
    assert rows == want["rows"] and cols == want["columns"]
    for ct in range(2):
        got = [xxhash.xxh64(np.ascontiguousarray(x).tobytes()).hexdigest() for x in engine.fetch_counts(ct)]
        assert got == want["ct%d" % ct], "cell type %d" % ct
