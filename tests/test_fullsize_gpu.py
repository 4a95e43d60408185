"""GPU: BASELINE's C2 and C4 workloads at their FULL sizes (10 M and 50 M reads) against digests the CPU oracle wrote.

tools/oracle_hashes.py evaluated the synthetic model on the host region by region (hostio.synth_records == the device generator,
tests/test_synth_gpu.py), counted every region's columns with oracle/count_oracle.c (lso_count_span_mt) on the build container's cores
and streamed the rows into xxhash digests (tests/golden/rows_hash_oracle_c2_10000000.json; 14 min on 6 threads).  Nothing the GPU wrote
is part of the pin.  The HIP rows are fetched cell type by cell type (4 GB of rows each) and hashed the same way."""
import json
import os

import numpy as np
import pytest
import xxhash

from longsom_amd import synth
from longsom_amd._lib import CountParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("cfg,n_reads", [("C2", 10_000_000), ("C4", 50_000_000)])
def test_full_size_rows_equal_the_cpu_oracle(engine, cfg, n_reads):
    """C2: 10 M reads x 5 k barcodes, 47.5 M rows.  C4: 50 M reads x 20 k barcodes, 48.0 M rows out of 4.6e10 events (71 min of the
    build container's CPU for the oracle; 124 GB of store on the device)."""
    from longsom_amd.engine import Engine
    engine.unload_reads()               # the session's handle gives its last load back: C4 needs 260 GB of the device's 288 while it is loaded
    want = json.load(open(os.path.join(G, "rows_hash_oracle_%s_%d.json" % (cfg.lower(), n_reads))))
    m = synth.named(cfg)
    assert m.n_reads == want["n_reads"] == n_reads
    p = CountParams.longsom_defaults()
    # twice, each in a handle of its own (C4 fills the device either way): the load that counts in its own pass and keeps no store
    # (lsg_set_store_policy: k_tm_count_direct, what bench.py times), then the load that builds the store, counted over it (k_tm_gather /
    # k_tm_resolve / k_tm_walk) - and, on the latter, the call digest
    # ... and first of all as bench.py's step makes it: the generator's events tile-phased (LSG_LAYOUT_PHASED), the entries binned by
    # 128-position windows, one 256-byte block per entry (k_tm_count_win)
    for how in ("count at load, no store, by windows", "count at load, no store", "store, then count"):
      m.layout = 1 if how.endswith("windows") else 0
      with Engine(0) as eng:
        eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2)
        eng.set_load_filter(p.min_mq, p.flag_exclude, p.ignore_orphans)          # as bench.py loads it
        if how.startswith("count"):
            eng.set_count_at_load(p); eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
        eng.synth_reads(m)
        eng.set_count_at_load(None); eng.set_store_policy(eng.STORE_KEEP)
        path = eng.layout_info()[0]
        # (C4's 64-position tiles hold more than max_depth = 200 000 reads; no position does - lsg_max_live_reads_exact -, so its load may count too)
        assert path == (6 if how.endswith("windows") else 4 if how.startswith("count") else 2), (how, eng.max_live_reads_all())
        rows, cols = eng.pileup_count(p)
        assert rows == want["rows"] and cols == want["columns"], how
        for ct in range(2):
            k, r, c = eng.fetch_counts(ct)
            got = [xxhash.xxh64(np.ascontiguousarray(x).tobytes()).hexdigest() for x in (k, r, c)]
            del k, r, c
            assert got == want["ct%d" % ct], "cell type %d (%s): rows of the full %s workload differ from the CPU oracle's" % (ct, how, cfg)
        calls_pin = os.path.join(G, "calls_hash_oracle_%s_%d.json" % (cfg.lower(), n_reads))
        if os.path.exists(calls_pin):
            # ... and merge + step 1 of the FULL workload: the step-1 text of every row step 2 keeps (7.69 M candidate rows of C2's 23.9 M merged
            # sites) hashes, contig by contig, to what the CPU oracles wrote (count_oracle.c + calling_oracle.py step 1 with scipy's betabinom
            # over 40 region shards: tools/oracle_call_hash.py C2 1e7 6 40, 27 min of the build container's CPU); nothing the GPU wrote is in the pin
            from tests.test_determinism_gpu import candidate_text_digest, device_text_digest, oracle_calls
            n_sites, n_cand = eng.call_step1()
            pin = json.load(open(calls_pin))
            assert n_sites == pin["merged_sites"] and n_cand == pin["candidate_rows"]
            assert candidate_text_digest(eng, m, candidates_only=True) == oracle_calls("%s_%d" % (cfg.lower(), n_reads))
            # ... and the text of the same rows as the DEVICE prints it (csrc/tables.hip; what the fused chain writes and step 2 reads)
            assert device_text_digest(eng, m) == oracle_calls("%s_%d" % (cfg.lower(), n_reads)), "device-printed step-1 rows (%s)" % how
