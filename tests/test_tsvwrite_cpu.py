"""The native TSV writers (csrc/hostio/tsvwrite.cpp) write byte for byte what the Python formatters of tsvio.py produce (those
are pinned to the reference's golden files in test_tsv_cpu.py / test_oracle_cpu.py)."""
import os

import numpy as np
import pytest

from longsom_amd import tsvio
from longsom_amd._lib import Call

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["chr1", "chr10", "chr2", "chrM"]                      # python string order != tid order


def random_counts(rng, n, n_contigs=4):
    keys = np.unique((rng.integers(0, n_contigs, n).astype(np.int64) << 32) | rng.integers(0, 5000, n))
    refs = rng.choice(np.frombuffer(b"ACGTN", np.uint8), len(keys))
    counts = rng.integers(0, 5, (len(keys), 42)).astype(np.uint32) * rng.integers(0, 2, (len(keys), 42)).astype(np.uint32)
    counts[:, 0] = rng.integers(1, 200000, len(keys)); counts[:, 1] = rng.integers(1, 5000, len(keys))
    return keys, refs, counts


def test_counts_and_merged_rows(tmp_path):
    rng = np.random.default_rng(1)
    per_ct = [random_counts(rng, 3000) for _ in range(3)]
    per_ct[1] = (per_ct[1][0], np.where(rng.random(len(per_ct[1][0])) < 0.1, ord("G"), per_ct[0][1][0]).astype(np.uint8), per_ct[1][2])   # REF conflicts
    date = "##fileDate=01/01/2000\n"
    p = str(tmp_path / "c.tsv")
    tsvio.write_counts_tsv(p, *per_ct[0], NAMES, "S.Cancer", date, threads=3)
    assert open(p).read() == tsvio.format_counts_tsv(*per_ct[0], NAMES, "S.Cancer", date)
    p = str(tmp_path / "m.tsv")
    head = tsvio.write_merged_tsv(p, per_ct, NAMES, ["A", "B", "C"], date, threads=3)
    want = tsvio.format_merged_tsv(per_ct, NAMES, ["A", "B", "C"], date)
    assert open(p).read() == want
    assert head == [l + "\n" for l in want.split("\n") if l.startswith("##")]
    # empty tables
    e = (np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros((0, 42), np.uint32))
    tsvio.write_merged_tsv(p, [e, e], NAMES, ["A", "B"], date)
    assert open(p).read() == tsvio.format_merged_tsv([e, e], NAMES, ["A", "B"], date)


def random_calls(rng, per_ct, n_ct):
    keys = np.unique(np.concatenate([k for k, _, _ in per_ct]))
    calls = np.zeros(len(keys), dtype=np.dtype(Call))
    idx = [dict(zip(k.tolist(), range(len(k)))) for k, _, _ in per_ct]
    for i, k in enumerate(keys.tolist()):
        c = calls[i]
        c["key"] = k; c["ref"] = ord("ACGT"[int(rng.integers(0, 4))])
        present = [k in idx[ct] for ct in range(n_ct)]
        cand = rng.random() < 0.4
        c["cell_types_min"] = int(rng.integers(0, n_ct + 1))
        c["sum_alts_bc"] = int(rng.choice([0, 0, 3, 40])); c["sum_dp"] = int(rng.integers(0, 5000)); c["sum_alts_cc"] = int(rng.integers(0, 30))
        c["sum_nc"] = int(rng.integers(-5, 900))
        c["noise_p_bc"] = int(rng.choice([-2, 0, 1, 37, 9999, 10000])); c["noise_p_cc"] = int(rng.choice([-2, 0, 5, 500, 10000]))
        if rng.random() < 0.8:
            up = rng.choice(np.frombuffer(b"ACGT", np.uint8), 5); dn = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(0, 6)))
            c["up_ctx"][:] = up; c["down_ctx"][:len(dn)] = dn
        sf = 0
        if cand and any(present):
            sf |= 1 << 31
            hc = 0
            for ct in range(n_ct):
                if present[ct] and rng.random() < 0.7:
                    hc |= 1 << ct
                    na = int(rng.integers(1, 4))
                    c["n_alt"][ct] = na
                    for q in range(na):
                        c["alt"][ct][q] = int(rng.integers(0, 4)); c["alt_bc"][ct][q] = int(rng.integers(1, 3000)); c["alt_cc"][ct][q] = int(rng.integers(1, 800))
                        c["p_bc"][ct][q] = int(rng.choice([0, 1, 12, 120, 5000, 10000])); c["p_cc"][ct][q] = int(rng.integers(0, 10001))
                    c["ct_filter"][ct] = int(rng.integers(1, 7))
            c["has_cand"] = hc
            sf |= int(rng.choice([0, 0, 1, 2, 4, 12, 32, 96]))
        elif rng.random() < 0.2:
            sf |= 16
        c["site_filter"] = sf
    return calls


@pytest.mark.parametrize("n_ct", [1, 2, 4])
def test_step1_rows(tmp_path, n_ct):
    rng = np.random.default_rng(10 + n_ct)
    per_ct = []
    for _ in range(n_ct):
        k, r, c = random_counts(rng, 2500)
        c[:, 0] = np.maximum(c[:, 0], 1); c[:, 1] = np.maximum(c[:, 1], 1)
        per_ct.append((k, r, c))
    calls = random_calls(rng, per_ct, n_ct)
    cts = ["T%d" % i for i in range(n_ct)]
    header = ["##fileDate=01/01/2000\n", "##INFO=x\n"]
    want = tsvio.format_step1_tsv(calls, per_ct, NAMES, cts, header)
    p = str(tmp_path / "s1.tsv")
    small = tsvio.write_step1_tsv(p, calls, per_ct, NAMES, cts, header, threads=4)
    assert open(p).read() == want
    # the small text = header + the rows step 2's filter keeps, in file order
    keep = [l for l in want.split("\n") if l and (l.startswith("#") or (l.split("\t")[4] != "." and l.split("\t")[5] != "."))]
    assert small == "\n".join(keep) + "\n"
    assert sum(1 for l in keep if not l.startswith("#")) > 100
    # the kept rows alone (nothing written, the other rows not formatted) and the table alone (nothing returned): what the fused chain uses
    assert tsvio.step1_kept_rows(calls, per_ct, NAMES, cts, header, threads=3, as_bytes=False) == small
    p2 = str(tmp_path / "s1b.tsv")
    assert tsvio.write_step1_tsv(p2, calls, per_ct, NAMES, cts, header, threads=4, collect=False) is None
    assert open(p2).read() == want


def test_ratio_and_p_text_match_python():
    """the two float texts of the step-1 table: str(round(a / float(b), 4)) and repr(k / 10000.0)"""
    rng = np.random.default_rng(5)
    k, r, c = random_counts(rng, 400)
    c[:, 0] = rng.integers(1, 100000, len(k)); c[:, 1] = rng.integers(1, 5000, len(k))
    calls = np.zeros(len(k), dtype=np.dtype(Call))
    calls["key"] = k; calls["ref"] = ord("A"); calls["site_filter"] = 1 << 31; calls["has_cand"] = 1; calls["n_alt"][:, 0] = 4; calls["ct_filter"][:, 0] = 6
    calls["alt"][:, 0, :] = [0, 1, 2, 3]
    calls["alt_bc"][:, 0, :] = rng.integers(0, 100000, (len(k), 4)); calls["alt_cc"][:, 0, :] = rng.integers(0, 5000, (len(k), 4))
    calls["alt_bc"][:3, 0, 0] = [1, 3, 5]; c[:3, 0] = [3, 8, 16]                      # 0.3333, 0.375, 0.3125 (exact ties in binary)
    calls["p_bc"][:, 0, :] = rng.integers(0, 10001, (len(k), 4)); calls["p_cc"][:, 0, :] = [0, 1, 10, 10000]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.tsv")
        tsvio.write_step1_tsv(p, calls, [(k, r, c)], NAMES, ["T"], [])
        assert open(p).read() == tsvio.format_step1_tsv(calls, [(k, r, c)], NAMES, ["T"], [])
