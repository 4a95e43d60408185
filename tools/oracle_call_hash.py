#!/usr/bin/env python3
"""Oracle pin of the CANDIDATE CALL RECORDS of a multi-million-read sample: the host evaluation of the synthetic model
(hostio.synth_records) counted by the CPU oracle (oracle/count_oracle.c), merged and called by oracle/calling_oracle.py (Python +
scipy.stats.betabinom, as the reference's BaseCellCalling.step1.py does), in `procs` processes over contiguous site ranges; the
xxhash of the text of the rows step 2 keeps (ALT != "." and FILTER != ".", step2.py:23) goes to tests/golden/.  Nothing the GPU wrote
is part of the pin; tests/test_determinism_gpu.py formats the HIP path's call records with the product's writer and compares.
usage: oracle_call_hash.py <config> <n_reads> [procs] [shards]
With shards > 1 the model is evaluated, counted and called region by region (longsom_amd.shard.region_shards, as tools/oracle_hashes.py
does for the count rows: every shard holds the reads overlapping its region and the oracle counts only the region's columns), the
digests are streamed: the FULL C2 workload (10 M reads, 23.9 M merged sites) never has to fit the host's memory."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longsom_amd import hostio, synth, tsvio  # noqa: E402
from oracle import calling_oracle, loader  # noqa: E402

STATE = {}


def work(span):
    lo, hi = span
    per_ct, names, fasta = STATE["per_ct"], STATE["names"], STATE["fasta"]
    sub = []
    for k, r, c in per_ct:
        a, b = np.searchsorted(k, lo), np.searchsorted(k, hi)
        sub.append((k[a:b], r[a:b], c[a:b]))
    merged = tsvio.format_merged_tsv(sub, names, ["Cancer", "Non-Cancer"])
    out = calling_oracle.step1(merged, fasta, info_lines=tsvio.STEP1_INFO_LINES)
    rows = []
    for line in out.split("\n"):
        if line and not line.startswith("#"):
            f = line.split("\t", 6)
            if f[4] != "." and f[5] != ".":
                rows.append(line)
    return len(rows), "".join(l + "\n" for l in rows)


def sharded(cfg, n, procs, shards):
    from longsom_amd.shard import region_shards, sub_model
    m = synth.named(cfg, n_reads=n)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    STATE.update(names=m.contig_names, fasta={m.contig_names[t]: refs[t].tobytes().decode() for t in range(len(refs))})
    h = xxhash.xxh64(); n_rows = 0; n_merged = 0
    per_contig = {}
    t0 = time.time()
    for i, (lo, hi, g_lo, g_hi) in enumerate(region_shards(m, shards)):
        rec = hostio.synth_records(sub_model(m, g_lo, g_hi))
        per_ct = [loader.count(rec, m.contig_len, refs, m.celltype_of, ct, threads=procs, span=(lo, hi))[:3] for ct in range(2)]
        del rec
        keys = np.unique(np.concatenate([p[0] for p in per_ct]))
        if not len(keys):
            continue
        n_merged += int(len(keys))
        n_chunks = procs * 4
        cuts = [int(keys[min(len(keys) - 1, len(keys) * j // n_chunks)]) for j in range(n_chunks)] + [int(keys[-1]) + 1]
        spans = [(cuts[j], cuts[j + 1]) for j in range(n_chunks) if cuts[j + 1] > cuts[j]]
        STATE["per_ct"] = per_ct
        with mp.get_context("fork").Pool(procs) as pool:           # (forked per shard: the workers see this shard's rows)
            for k, text in pool.imap(work, spans):
                h.update(text.encode()); n_rows += k
                for line in text.split("\n"):
                    if line:
                        c = line[:line.index("\t")]
                        e = per_contig.setdefault(c, [0, xxhash.xxh64()])
                        e[0] += 1; e[1].update((line + "\n").encode())
        print("shard %d/%d: %d merged sites, %d candidate rows so far (%.0f s)" % (i + 1, shards, n_merged, n_rows, time.time() - t0), flush=True)
    out = {"config": cfg, "n_reads": n, "merged_sites": n_merged, "candidate_rows": n_rows, "xxh64_of_candidate_row_text": h.hexdigest(),
           "order": "contigs in BAM header (tid) order, positions ascending", "per_contig": {c: [e[0], e[1].hexdigest()] for c, e in per_contig.items()},
           "source": "oracle/count_oracle.c + oracle/calling_oracle.py step1 (scipy betabinom) over hostio.synth_records, %d region shards, %d processes (CPU)" % (shards, procs)}
    path = os.path.join(ROOT, "tests", "golden", "calls_hash_oracle_%s_%d.json" % (cfg.lower(), n))
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: v for k, v in out.items() if k != "per_contig"})


def main():
    cfg, n = sys.argv[1], int(float(sys.argv[2]))
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else os.cpu_count()
    if len(sys.argv) > 4 and int(sys.argv[4]) > 1:
        return sharded(cfg, n, procs, int(sys.argv[4]))
    m = synth.named(cfg, n_reads=n)
    t0 = time.time()
    rec = hostio.synth_records(m)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    per_ct = [loader.count(rec, m.contig_len, refs, m.celltype_of, ct, threads=procs)[:3] for ct in range(2)]
    del rec
    print("counted: rows %s (%.0f s)" % ([len(p[0]) for p in per_ct], time.time() - t0), flush=True)
    keys = np.unique(np.concatenate([p[0] for p in per_ct]))
    n_chunks = procs * 24
    cuts = [int(keys[min(len(keys) - 1, len(keys) * i // n_chunks)]) for i in range(n_chunks)] + [int(keys[-1]) + 1]
    spans = [(cuts[i], cuts[i + 1]) for i in range(n_chunks) if cuts[i + 1] > cuts[i]]
    STATE.update(per_ct=per_ct, names=m.contig_names, fasta={m.contig_names[t]: refs[t].tobytes().decode() for t in range(len(refs))})
    h = xxhash.xxh64(); n_rows = 0
    per_contig = {}                                  # contig -> [rows, xxh64 of its rows' text]: a mismatch names the contig
    with mp.get_context("fork").Pool(procs) as pool:
        for i, (k, text) in enumerate(pool.imap(work, spans)):
            h.update(text.encode()); n_rows += k
            for line in text.split("\n"):
                if line:
                    c = line[:line.index("\t")]
                    e = per_contig.setdefault(c, [0, xxhash.xxh64()])
                    e[0] += 1; e[1].update((line + "\n").encode())
            if i % procs == 0:
                print("chunk %d/%d, %d candidate rows so far (%.0f s)" % (i + 1, len(spans), n_rows, time.time() - t0), flush=True)
    out = {"config": cfg, "n_reads": n, "merged_sites": int(len(keys)), "candidate_rows": n_rows, "xxh64_of_candidate_row_text": h.hexdigest(),
           "order": "contigs in BAM header (tid) order, positions ascending", "per_contig": {c: [e[0], e[1].hexdigest()] for c, e in per_contig.items()},
           "source": "oracle/count_oracle.c + oracle/calling_oracle.py step1 (scipy betabinom) over hostio.synth_records, %d processes (CPU)" % procs}
    path = os.path.join(ROOT, "tests", "golden", "calls_hash_oracle_%s_%d.json" % (cfg.lower(), n))
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, out)


if __name__ == "__main__":
    main()
