#!/usr/bin/env python3
"""Oracle pins for the multi-million-read samples: the HOST evaluation of the synthetic model (hostio.synth_records, bit-identical
to the device generator: tests/test_synth_gpu.py) counted by the region-parallel CPU oracle (oracle/count_oracle.c lso_count_mt),
hashed the way tests/test_determinism_gpu.py hashes the HIP path's rows.  Runs on the CPU (build container); the JSON it writes
under tests/golden/ is what the GPU rows must reproduce.   usage: oracle_hashes.py <config> <n_reads> [threads] [shards]
With shards > 1 the model is evaluated and counted region by region (longsom_amd.shard.region_shards: every shard holds the reads that
overlap its region, the oracle counts only the region's columns, lso_count_span_mt) and the digests are streamed — the full-size
workloads (C2 at 10 M reads: 1.2e10 events) never have to fit the host's memory at once; the digests equal the one-piece ones."""
import json
import os
import sys
import time

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longsom_amd import hostio, synth  # noqa: E402
from oracle import loader  # noqa: E402


def sharded(cfg, n, threads, shards):
    from longsom_amd.shard import region_shards, sub_model
    m = synth.named(cfg, n_reads=n)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    hs = [[xxhash.xxh64() for _ in range(3)] for _ in range(2)]
    out = {"config": cfg, "n_reads": n, "source": "oracle/count_oracle.c lso_count_span_mt over hostio.synth_records, %d region shards (CPU)" % shards,
           "rows": [0, 0], "columns": 0}
    t_all = time.time()
    for i, (lo, hi, g_lo, g_hi) in enumerate(region_shards(m, shards)):
        t0 = time.time()
        rec = hostio.synth_records(sub_model(m, g_lo, g_hi))
        for ct in range(2):
            k, r, c, ncol = loader.count(rec, m.contig_len, refs, m.celltype_of, ct, threads=threads, span=(lo, hi))
            for h, x in zip(hs[ct], (k, r, c)):
                h.update(np.ascontiguousarray(x).tobytes())
            out["rows"][ct] += int(len(k)); out["columns"] += int(ncol)
        print("shard %d/%d: %d reads %d events, rows so far %s, %.1f s (%.0f s in all)" % (i + 1, shards, rec.n_reads, rec.n_events, out["rows"], time.time() - t0, time.time() - t_all), flush=True)
        del rec
    for ct in range(2):
        out["ct%d" % ct] = [h.hexdigest() for h in hs[ct]]
    return out


def main():
    cfg, n = sys.argv[1], int(float(sys.argv[2]))
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else os.cpu_count()
    shards = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    if shards > 1:
        out = sharded(cfg, n, threads, shards)
        path = os.path.join(ROOT, "tests", "golden", "rows_hash_oracle_%s_%d.json" % (cfg.lower(), n))
        json.dump(out, open(path, "w"), indent=1)
        print("wrote", path)
        return
    m = synth.named(cfg, n_reads=n)
    t0 = time.time()
    rec = hostio.synth_records(m)
    refs = [hostio.ref_bases(m.seed, t, int(l)) for t, l in enumerate(m.contig_len)]
    print("model: %d reads %d segments %d events (%.1f s)" % (rec.n_reads, rec.n_segs, rec.n_events, time.time() - t0), flush=True)
    out = {"config": cfg, "n_reads": n, "source": "oracle/count_oracle.c lso_count_mt over hostio.synth_records (CPU)", "rows": [], "columns": 0}
    for ct in range(2):
        t0 = time.time()
        k, r, c, ncol = loader.count(rec, m.contig_len, refs, m.celltype_of, ct, threads=threads)
        out["ct%d" % ct] = [xxhash.xxh64(np.ascontiguousarray(x).tobytes()).hexdigest() for x in (k, r, c)]
        out["rows"].append(int(len(k))); out["columns"] += int(ncol)
        print("ct %d: %d rows, %d columns, %.1f s on %d threads" % (ct, len(k), ncol, time.time() - t0, threads), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "rows_hash_oracle_%s_%d.json" % (cfg.lower(), n))
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
