"""How far do the step-1 p-values of a workload lie from a 4-decimal rounding tie?  The text of BaseCellCalling.step1 prints
round(betabinom.sf(k - 0.1, n, alpha, beta), 4) (step1.py:196,201) and round(1 - betabinom.cdf(...), 4) (:329-330); the device's
fp64 tail and scipy's differ by ~1e-13, so the printed digits can only differ where the true p lies that close to a tie (x.xxxx5).
This tool takes every distinct (k, n) the candidates of a sample were tested with — read counts against (alpha1, beta1), cell
counts against (alpha2, beta2), the Rest_BC / Rest_CC noise tests — evaluates the unrounded tail on the device (lsg_betabinom_sf),
prints the histogram of the distance to the nearest tie, and re-evaluates everything closer than `--near` with scipy itself,
reporting any pair whose printed value differs.   usage: python tools/p_margins.py [config] [n_reads] [--near 1e-7]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from longsom_amd import synth  # noqa: E402
from longsom_amd._lib import CallParams  # noqa: E402
from longsom_amd.engine import Engine  # noqa: E402


def pairs_of(calls, per_ct):
    """distinct (k, n) per parameter pair: [reads], [cells]"""
    key = calls["key"]
    reads, cells = [], []
    for ct, (k, _, c) in enumerate(per_ct):
        at = np.searchsorted(k, key)
        ok = (at < len(k))
        ok[ok] = k[at[ok]] == key[ok]
        dp = np.zeros(len(key), np.int64); nc = np.zeros(len(key), np.int64)
        dp[ok] = c[at[ok], 0]; nc[ok] = c[at[ok], 1]
        for q in range(calls["alt_bc"].shape[2]):
            use = ok & (calls["n_alt"][:, ct] > q)
            reads.append(np.stack([calls["alt_bc"][use, ct, q].astype(np.int64), dp[use]], 1))
            cells.append(np.stack([calls["alt_cc"][use, ct, q].astype(np.int64), nc[use]], 1))
    noise = calls["sum_alts_bc"] > 0
    reads.append(np.stack([calls["sum_alts_bc"][noise].astype(np.int64), calls["sum_dp"][noise].astype(np.int64)], 1))
    ok2 = noise & (calls["sum_nc"] >= 0) & (calls["sum_alts_cc"] >= 0)
    cells.append(np.stack([calls["sum_alts_cc"][ok2].astype(np.int64), calls["sum_nc"][ok2].astype(np.int64)], 1))
    u = lambda x: np.unique(np.concatenate(x), axis=0)
    r, c = u(reads), u(cells)
    return r[(r[:, 0] >= 1) & (r[:, 0] <= r[:, 1])], c[(c[:, 0] >= 1) & (c[:, 0] <= c[:, 1])]


def margins(eng, pairs, alpha, beta):
    p4, p = eng.betabinom_sf(pairs[:, 0], pairs[:, 1], alpha, beta)
    x = p * 1e4
    dist = np.abs(x - np.floor(x) - 0.5) / 1e4          # distance of p to the nearest value that rounds either way
    return p4, p, dist


def audit(eng, calls, per_ct, params, near=1e-7):
    from scipy.stats import betabinom
    out = {}
    for name, pairs, (a, b) in (("reads", None, (params.alpha1, params.beta1)), ("cells", None, (params.alpha2, params.beta2))):
        pairs = pairs_of(calls, per_ct)[0 if name == "reads" else 1]
        p4, p, dist = margins(eng, pairs, a, b)
        edges = [0, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1]
        hist = np.histogram(dist, bins=edges)[0].tolist()
        close = np.flatnonzero(dist < near)
        flips = []
        for i in close.tolist():
            k, n = int(pairs[i, 0]), int(pairs[i, 1])
            want = round(float(betabinom.sf(k - 0.1, n, a, b)), 4)
            if abs(want * 1e4 - int(p4[i])) > 0.5:
                flips.append([k, n, float(p[i]), int(p4[i]), want])
        out[name] = {"pairs": int(len(pairs)), "min_distance": float(dist.min()) if len(dist) else None, "histogram_edges": edges, "histogram": hist,
                     "within_near": int(len(close)), "near": near, "differ_from_scipy": flips}
    return out


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    near = float(sys.argv[sys.argv.index("--near") + 1]) if "--near" in sys.argv else 1e-7
    m = synth.named(cfg, n_reads=n)
    eng = Engine(0)
    eng.set_contigs(m.contig_len); eng.synth_reference(m.seed); eng.set_barcodes(m.celltype_of, 2); eng.synth_reads(m)
    eng.pileup_count(); eng.call_step1()
    calls = eng.fetch_calls(candidates_only=True)
    per_ct = [eng.fetch_counts(ct) for ct in range(2)]
    res = audit(eng, calls, per_ct, CallParams.longsom_defaults(), near)
    res["workload"] = "%s at %d reads: %d candidate / noisy rows" % (cfg, n, len(calls))
    print(json.dumps(res, indent=1))
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/p_margins_%s_%d.json" % (cfg.lower(), n), "w"), indent=1)
