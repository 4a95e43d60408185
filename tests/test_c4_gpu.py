"""GPU: BASELINE.json's configuration C4 as specified — the deep-coverage workload WITH its position sets resident in HBM: 5 M
panel-of-normals positions and 15 M RNA-editing positions (lsg_load_posset), every step-1 candidate of the sample probed against
them (GetExtraFilters' membership tests, BaseCellCalling.step2.py:142-158; build_dict's sets :197-221), hits compared with numpy.
The sample is C4's model at 2.5 M reads (its count rows are pinned to the CPU oracle in tests/test_determinism_gpu.py); the full
50 M-read pass with the same sets is run by tools/c4_run.py and recorded under profiles/."""
import time

import numpy as np
import pytest

from longsom_amd import synth
from tests.support import possets

pytestmark = pytest.mark.gpu


def test_c4_candidates_against_resident_position_sets(engine):
    m = synth.named("C4", n_reads=2_500_000)
    engine.set_contigs(m.contig_len); engine.synth_reference(m.seed); engine.set_barcodes(m.celltype_of, 2)
    engine.set_region()
    engine.synth_reads(m)
    engine.pileup_count()
    n_sites, n_cand = engine.call_step1()
    calls = engine.fetch_calls(candidates_only=True)
    q = np.ascontiguousarray(calls["key"] + 1)                           # (tid << 32) | 1-based position, as step 2 builds its queries
    assert len(q) >= n_cand > 2_000_000
    sets = {0: possets.random_keys(40, possets.C4_SIZES["editing"], m.contig_len, salt=q, salt_frac=0.01),
            1: possets.random_keys(41, possets.C4_SIZES["pon"], m.contig_len, salt=q, salt_frac=0.03),
            2: possets.random_keys(42, 50_000, m.contig_len, salt=q, salt_frac=0.001)}
    assert len(sets[0]) > 14_000_000 and len(sets[1]) > 4_800_000
    for kind, keys in sets.items():
        engine.load_posset(kind, keys)                                   # 160 MB of keys stay resident beside the reads
    for kind, keys in sets.items():
        t0 = time.perf_counter()
        hits = engine.probe_posset(kind, q)
        dt = time.perf_counter() - t0
        want = np.zeros(len(q), np.uint8)
        at = np.searchsorted(keys, q)
        ok = at < len(keys)
        want[ok] = (keys[at[ok]] == q[ok])
        assert np.array_equal(hits, want), "set %d" % kind
        assert 0 < int(hits.sum()) < len(q)
        print("set %d: %d keys, %d queries, %d hits, probe incl. copies %.1f ms" % (kind, len(keys), len(q), int(want.sum()), dt * 1e3))
    # the count path is unaffected by the resident sets
    rows2, _ = engine.pileup_count()
    assert sum(rows2) > 5_000_000
