"""Development aid (GPU box): count a small random case and name the rows and fields that differ from the CPU oracle
(used when the hand-written per-entry body of k_tm_walk is touched)."""
import os
import sys, numpy as np
sys.path.insert(0, '.')
from tests.test_count_gpu import make_case
from longsom_amd._lib import CountParams
from longsom_amd.engine import Engine
from oracle import loader
eng = Engine(0)
lens = [5000, 1200, 70]
rec, refs, ct_of = make_case(1, 3000, lens, 50)
params = CountParams.longsom_defaults()
eng.set_contigs(lens)
for t, r in enumerate(refs): eng.load_reference(t, r)
eng.set_barcodes(ct_of, 2); eng.load_reads(rec)
n_rows, n_cols = eng.pileup_count(params)
names = ["DP","NC"]+["CC%d"%i for i in range(8)]+["BC%d"%i for i in range(8)]+["BQ%d"%i for i in range(8)]+["BCf%d"%i for i in range(8)]+["BCr%d"%i for i in range(8)]
for ct in range(2):
    k, rf, c = eng.fetch_counts(ct)
    ok, orf, oc, ocols = loader.count(rec, lens, refs, ct_of, ct, params.min_bq, params.min_mq, params.min_dp, params.min_cc, params.flag_exclude, params.ignore_orphans)
    print("ct", ct, "rows", n_rows[ct], len(ok), "keys equal", len(k)==len(ok) and bool((k==ok).all()))
    if len(k)==len(ok):
        d = (c != oc)
        print(" differing cells:", int(d.sum()), "rows:", int(d.any(1).sum()))
        cols = np.flatnonzero(d.any(0)); print(" cols:", [names[j] if j < len(names) else j for j in cols])
        for r in np.flatnonzero(d.any(1))[:6]:
            js = np.flatnonzero(d[r]); print("  row", r, [(names[j], int(c[r,j]), int(oc[r,j])) for j in js])
