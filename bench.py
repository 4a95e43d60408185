#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its named configuration.

  metric   genomic sites/s, pileup + call: pileup columns with >= 1 counted entry, summed over cell
           types, per second of one full ONE-SHOT pass over a BAM's worth of decoded reads, as a LongSom rule runs it
           (one count per BAM, BaseCellCounter.py:182-320):
             lsg_load_reads on device-resident compact read-record arrays (the device half of ingest: every (segment x 64-position
             tile) entry binned per tile and sorted by barcode, and - lsg_set_count_at_load + lsg_set_store_policy - the BAM's one
             count made in the same pass straight from the caller's events: a BAM that is counted once keeps no tile store, and - its reads
             loaded under the count's own read filters, lsg_set_load_filter - carries 8-byte keys alone through the scatter and the sort)
             -> lsg_pileup_count (hands that count over) -> lsg_call_step1 (merge + step 1) [-> all-gather of PASS-candidate call
             rows when N > 1]
           A step = one such pass on a FRESH load; the timed steps follow a warm-up load.  LSG_BENCH_KEEP_STORE=1: the load also writes
           the tile store later counts would work on (k_tm_gather_count); LSG_BENCH_TWO_PASS=1: load and count apart (round 3's step).
  workload C2 (BASELINE.json configs[1]): whole-genome synthetic long-read workload, 10 M reads x 5 k
           barcodes, generated directly in HBM by the model of longsom_amd/csrc/synth_model.h
           (inputs are resident when the timed region starts: the arrays lsg_synth_generate left in HBM)
  N > 1    strong scaling: the same 10 M-read workload, genomic windows sharded over the ranks by
           estimated work; every rank loads the reads overlapping its region, counts only its own
           columns and the ranks all-gather their PASS-candidate call rows over RCCL.

One JSON line on rank 0.  `roofline` is for the kernel that takes the most time of a step - k_tm_count_direct, the count made inside the
load (one 2-byte load per entry and lane from the caller's events, counters in LDS) - priced by SURVEY 8(d)'s bytes alone: 2 B per
admitted event + 24 B per admitted read + 168 B per row it emits (sort scratch and re-reads are NOT algorithmic), over its HIP-event
time on its own stream.  `roofline.path_frac` prices the WHOLE step the same way: (2 E + 24 R + 168 S_emit) / ms_per_step / peak.
`roofline.step_traffic` is the fabric traffic of every kernel of a step from the newest profiles/ file recorded for this build (sum and
its ratio to the algorithmic bytes).  `config.recount_ms` is a count + call pass over a RESIDENT store (what the second pass of the
re-annotation loop pays), measured after the clock on one extra load that keeps its store (`config.load_that_also_writes_the_store_ms`).
`cpu_baseline` times the CPU oracle (oracle/, kind "port") on a bounded sample of the same workload on ALL of this box's host cores.
`roofline.traffic` is quoted from the newest profiles/ file only when that file was recorded for this build of the kernels (else null +
traffic_stale).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process has touched neither torch nor HIP yet,
    so it starts the N ranks itself — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...` as a CHILD process
    (never an exec) — relays rank 0's JSON line and exits with the child's code.  The fan-out this replaces is the reference's
    mp.Pool(CORE) over windows (BaseCellCounter.py:392-402)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line); sys.stdout.flush()
    return proc.wait()

METRIC = "genomic sites/s pileup+call, 10M-read BAM x 5k barcodes, 1/2/4/8 MI355X"
CALL_BYTES = 336          # sizeof(lsg_call)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec



def host_cores(cap=16):
    """host threads this process may really use: the scheduler affinity, the cgroup CPU quota, and the GPU box's share per GPU
    (16; os.cpu_count() reports the whole machine there)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = int(txt[0]) if txt[0] != "max" else -1
            period = int(txt[1]) if len(txt) > 1 else int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, int(os.environ.get("LSG_HOST_CORES", cap))))


_CALL_STATE = {}


def _call_span(span):
    """one process of the CPU baseline's step 1: merge + calling_oracle.step1 over the sites in [lo, hi) (forked: the rows are inherited)"""
    from oracle import calling_oracle
    from longsom_amd import tsvio
    lo, hi = span
    sub = []
    for k, r, c in _CALL_STATE["per_ct"]:
        a, b = np.searchsorted(k, lo), np.searchsorted(k, hi)
        sub.append((k[a:b], r[a:b], c[a:b]))
    merged = tsvio.format_merged_tsv(sub, _CALL_STATE["names"], ["Cancer", "Non-Cancer"])
    out = calling_oracle.step1(merged, _CALL_STATE["fasta"], info_lines=tsvio.STEP1_INFO_LINES)
    return sum(1 for l in out.split("\n") if l and not l.startswith("#"))


def cpu_baseline(eng, model, target_reads=150_000):
    """The CPU side of BASELINE.md §3 on this box's host cores, bounded to ~20 s:
      cpu_native  oracle/count_oracle.c (region-parallel, ALL host cores) on a contiguous-gene sample of the C2 workload, plus the
                  step-1 oracle (Python + scipy, one process, as the reference's step 1 is) scaled from a bounded number of sites;
                  the same count on ONE core on a third of the sample beside it.  `value` is the all-cores figure.
      cpu_pyloop  oracle/pyloop.py: the reference's per-read Python loop structure (BaseCellCounter.py:198-312) on 50 kb windows of
                  the C1 workload, one core and a process pool over all cores (the reference's mp.Pool, :392-402).
    Also checks the GPU rows of the sample against the oracle (a parity check at bench time)."""
    from oracle import calling_oracle, loader, pyloop
    from longsom_amd import hostio, tsvio
    cores = host_cores()
    reads = np.diff(model.gene_read_off)
    g_lo = model.n_genes // 3
    g_hi = g_lo
    while g_hi < model.n_genes and int(reads[g_lo:g_hi].sum()) < target_reads:
        g_hi += 1
    sm = sub_model(model, g_lo, g_hi)
    eng.set_region()
    eng.set_keep_reads(True)                 # the sample's arrays come back to the host for the oracle
    eng.synth_reads(sm)
    eng.set_keep_reads(False)
    rows, cols = eng.pileup_count()
    rec = eng.reads_to_host()
    tids = sorted(set(sm.gene_tid.tolist()))
    refs = [eng.reference_to_host(t) if t in tids else np.zeros(0, np.uint8) for t in range(len(model.contig_len))]
    t0 = time.time()
    ok, per_ct, n_cols = True, [], 0
    for ct in range(2):
        k, rf, c, ncol = loader.count(rec, model.contig_len, refs, model.celltype_of, ct, threads=cores)
        per_ct.append((k, rf, c)); n_cols += ncol
    t_count = time.time() - t0
    for ct in range(2):
        gk, gr, gc = eng.fetch_counts(ct)
        ok &= bool(np.array_equal(gk, per_ct[ct][0]) and np.array_equal(gc, per_ct[ct][2]))
    ok &= n_cols == cols
    # one core, a third of the sample (the single-threaded form of the same oracle)
    third = rec.subset(np.arange(rec.n_reads) < rec.n_reads // 3)
    t0 = time.time()
    cols1 = sum(loader.count(third, model.contig_len, refs, model.celltype_of, ct)[3] for ct in range(2))
    t_count1 = time.time() - t0
    # step 1 of EVERY merged site of the sample (scipy betabinom per alt, as the reference does), measured: `cores` processes, each over a
    # contiguous range of the sites (the reference's step 1 is one process; an all-cores baseline gives it the same cores as the count)
    import multiprocessing as mp
    keys = np.unique(np.concatenate([p[0] for p in per_ct])) if n_cols else np.zeros(0, np.int64)
    n_merged = int(len(keys))
    fasta = {model.contig_names[t]: refs[t].tobytes().decode() for t in tids}
    _CALL_STATE.update(per_ct=per_ct, names=model.contig_names, fasta=fasta)
    n_chunks = max(1, cores * 4)
    cuts = [int(keys[min(n_merged - 1, n_merged * i // n_chunks)]) for i in range(n_chunks)] + [int(keys[-1]) + 1] if n_merged else [0, 0]
    spans = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        n_call = sum(pool.imap_unordered(_call_span, spans))
    t_call = time.time() - t0
    assert n_call == n_merged, "the step-1 oracle's processes did not cover the sample's merged sites"
    t_call_site = t_call * cores / max(1, n_call)                   # CPU time per site
    t_total = t_count + t_call
    # cpu_pyloop on C1 (chr22-only, 50 k reads, 200 barcodes): the reference's loop structure on its 50 kb windows
    c1 = synth.named("C1")
    rec1 = hostio.synth_records(c1)
    refs1 = [hostio.ref_bases(c1.seed, t, int(l)) for t, l in enumerate(c1.contig_len)]
    bcs = hostio.synth_barcodes(c1)
    span_lo = int(rec1.seg_start.min()) // 50000 * 50000 + 1
    wins = [(0, x, x + 50000) for x in range(span_lo, int(c1.contig_len[0]), 50000)]
    seg_tile = (rec1.seg_start // 50000)
    busy = sorted(wins, key=lambda w: -int(((seg_tile == (w[1] // 50000))).sum()))      # the busiest windows first
    adm = [pyloop.admitted_reads(rec1, c1.celltype_of, ct) for ct in range(2)]

    def cols_of(jobs):
        return sum(sum(1 for _ in pyloop.columns_of(rec1, tid, lo, hi, 20, adm[ct])) for ct, tid, lo, hi in jobs)
    jobs1 = [(ct, *w) for w in busy[:1] for ct in range(2)]
    t0 = time.time(); rows_py1 = pyloop.count_windows(rec1, bcs, refs1, c1.contig_names, c1.celltype_of, jobs1, 1); t_py1 = time.time() - t0
    cols_py1 = cols_of(jobs1)
    jobsn = [(ct, *w) for w in busy[:max(1, min(len(busy), cores))] for ct in range(2)]
    t0 = time.time(); rows_pyn = pyloop.count_windows(rec1, bcs, refs1, c1.contig_names, c1.celltype_of, jobsn, cores); t_pyn = time.time() - t0
    cols_pyn = cols_of(jobsn)
    return {"value": n_cols / t_total if t_total > 0 else 0.0, "unit": "sites/s", "cores": cores, "kind": "port",
            "count_only": {"unit": "sites/s", "all_cores": n_cols / t_count if t_count > 0 else 0.0, "one_core": cols1 / t_count1 if t_count1 > 0 else 0.0,
                           "what": "oracle/count_oracle.c alone (lso_count_mt on %d threads / lso_count on one), no step 1" % cores},
            "step1_only": {"unit": "merged sites/s", "all_cores": n_call / t_call if t_call > 0 else 0.0, "per_process": 1.0 / t_call_site if t_call_site > 0 else 0.0,
                           "what": "oracle/calling_oracle.py step1 (Python + scipy.stats.betabinom, as BaseCellCalling.step1.py:196-201), %d processes over contiguous site ranges" % cores},
            "sample": "cpu_native: %d reads of %d contiguous genes of the C2 workload (%d events, %d columns), both stages MEASURED on %d cores: oracle/count_oracle.c "
                      "(lso_count_mt, %d threads, %.1f s) + merge and oracle/calling_oracle.py step1 of all %d merged sites in %d processes (%.1f s)"
                      % (rec.n_reads, g_hi - g_lo, rec.n_events, n_cols, cores, cores, t_count, n_merged, cores, t_call),
            "native_1core": {"value": cols1 / (t_count1 + t_call_site * n_merged / 3) if t_count1 > 0 else 0.0, "unit": "sites/s",
                             "sample": "%d reads (a third of the sample), lso_count single-threaded %.1f s" % (third.n_reads, t_count1)},
            "pyloop": {"kind": "port", "unit": "sites/s", "workload": "C1 (chr22-only, 50 k reads, 200 barcodes), the busiest 50 kb windows x 2 cell types",
                       "value_1core": cols_py1 / t_py1 if t_py1 > 0 else 0.0, "value_allcores": cols_pyn / t_pyn if t_pyn > 0 else 0.0, "cores": cores,
                       "sample": "1 core: %d jobs, %d columns, %d rows, %.1f s; pool of %d: %d jobs, %d columns, %d rows, %.1f s" % (len(jobs1), cols_py1, rows_py1, t_py1, cores, len(jobsn), cols_pyn, rows_pyn, t_pyn),
                       "note": "per-read Python loop with the structure of BaseCellCounter.py:198-312 on columns cut from the decoded arrays; pysam's per-read object "
                               "construction is not included, so this is an upper bound on the reference's own columns/s"},
            "gpu_matches_oracle_on_sample": ok}


def end_to_end(n_reads):
    """files in -> every output file (pipeline.run_snv: ingest, count, merge, steps 1-3, all tables written), measured in THIS run on a
    BAM written in this run: C2's model at n_reads reads (default: all 10 M of the metric's BAM).  The count rows that come back from
    the device on their way to the tables are hashed and compared with the digests the CPU oracle wrote for this workload
    (tests/golden/rows_hash_oracle_c2_<reads>.json: tools/oracle_hashes.py; calls_hash_oracle_*.json's totals): `tables_match_oracle`.
    After the timed steps, on rank 0 at N=1; never part of `value`."""
    import shutil
    import tempfile
    from longsom_amd import hostio, pipeline
    d = tempfile.mkdtemp(prefix="lsg_bench_e2e_")
    try:
        m = synth.named("C2", n_reads=n_reads)
        bam, fa, bct = os.path.join(d, "S.bam"), os.path.join(d, "ref.fa"), os.path.join(d, "bc.tsv")
        t0 = time.time(); hostio.synth_bam(m, bam, fa); t_bam = time.time() - t0
        hostio.write_barcodes_tsv(bct, hostio.synth_barcodes(m), m.celltype_of, ["Cancer", "Non-Cancer"])
        t0 = time.time()
        out = pipeline.run_snv(bam, bct, fa, os.path.join(d, "out"), "S", params=pipeline.SnvParams(row_digests=True))
        wall = time.time() - t0 - float(out.timings.get("row_digests", 0.0))          # (the check itself is not part of the run)
        mb = lambda ps: round(sum(os.path.getsize(q) for q in ps) / 1e6, 1)
        match, pin_of = None, None
        pin = os.path.join(ROOT, "tests", "golden", "rows_hash_oracle_c2_%d.json" % n_reads)
        if os.path.exists(pin) and out.row_digests:
            want = json.load(open(pin)); got = out.row_digests
            match = got["rows"] == want["rows"] and got["columns"] == want["columns"] and all(got["ct%d" % ct] == want["ct%d" % ct] for ct in range(2))
            pin_of = os.path.relpath(pin, ROOT)
            cpin = os.path.join(ROOT, "tests", "golden", "calls_hash_oracle_c2_%d.json" % n_reads)
            if os.path.exists(cpin):
                cw = json.load(open(cpin))
                match = match and got["merged_sites"] == cw["merged_sites"] and got["candidate_rows"] == cw["candidate_rows"]
                pin_of += " + " + os.path.relpath(cpin, ROOT) + " (merged sites, candidate rows)"
        return {"measured": "in this run", "tables_match_oracle": match, "oracle_pin": pin_of, "workload": "C2's model at %d reads x %d barcodes as a BAM of %.0f MB + FASTA + barcodes.tsv, written in this run (%.1f s, not counted); "
                                                       "BAM -> per-cell-type count tables, merged table, step-1/2/3 tables on disk" % (n_reads, m.n_cb, os.path.getsize(bam) / 1e6, t_bam),
                "wall_s": round(wall, 2), "seconds": {k: round(float(v), 3) for k, v in out.timings.items()},
                "out_MB": {"counts": mb(out.counts.values()), "merged": mb([out.merged]), "step1": mb([out.step1])},
                "step3_rows": max(0, sum(1 for l in open(out.step3) if not l.startswith("#")) - 1), "host_threads": os.cpu_count()}
    except Exception as e:                                           # the contract line is still printed; the failure is in it and on stderr
        print("bench.py: end-to-end leg failed: %r" % (e,), file=sys.stderr)
        return {"measured": "failed", "error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def c4_oneshot(device_index, steps=3):
    """BASELINE.json configs[3] (50 M reads x 20 k barcodes, 5.3e10 events: 125 GB of tile-phased events in HBM) as ONE-SHOT steps like the
    timed ones: generated on the device (untimed, ~4 s), then load (windows, keys alone, no store) + count + merge + step-1 call per step,
    the first one a warm-up (allocations, the tail table).  After the clock, N = 1 only; never part of `value`.  Its rows equal the CPU
    oracle's at full size in tests/test_fullsize_gpu.py."""
    import torch
    try:
        torch.cuda.synchronize()
        if torch.cuda.mem_get_info(device_index)[0] < 250e9:
            return {"measured": "skipped", "why": "less than 250 GB of the device free"}
        m = synth.named("C4", layout=0 if os.environ.get("LSG_BENCH_COMPACT") == "1" else 1)
        cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
        with Engine(device_index) as e4:
            e4.set_contigs(m.contig_len); e4.synth_reference(m.seed); e4.set_barcodes(m.celltype_of, 2)
            e4.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)
            e4.set_count_at_load(cp); e4.set_store_policy(e4.STORE_SKIP_WHEN_COUNTED)
            t0 = time.time(); reads = e4.synth_generate(m); t_gen = time.time() - t0
            best = None
            for i in range(steps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                e4.load_reads_struct(reads)
                rows, cols = e4.pileup_count(cp)
                n_sites, n_cand = e4.call_step1(kp)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
                st = e4.count_stats()
                bytes_k = 2.0 * st.n_events_admitted + 24.0 * st.n_reads_admitted + 168.0 * st.rows_by_kernel[1]
                bytes_p = 2.0 * st.n_events_admitted + 24.0 * st.n_reads_admitted + 168.0 * sum(rows)
                cur = {"ms_per_step": round(dt, 2), "count_kernel_ms": round(float(st.ms_walk), 2), "kernel_frac": round(bytes_k / max(st.ms_walk, 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                       "path_frac": round(bytes_p / dt / 1e6 / HBM_PEAK_GBS, 4), "sites_per_s": cols / (dt / 1e3), "load_path": e4.layout_info()[0],
                       "build_ms": [round(float(x), 2) for x in e4.build_times()]}
                if i and (best is None or cur["ms_per_step"] < best["ms_per_step"]):
                    best = cur
            best.update({"measured": "in this run", "workload": "C4: 50 M reads x 20 k barcodes, %d events in a %.1f GB event array, one-shot steps (best of %d after a warm-up)" % (int(st.n_events_admitted), 2 * int(reads.n_events) / 1e9, steps - 1),
                         "sites_counted": int(cols), "rows_emitted": int(sum(rows)), "merged_sites": int(n_sites), "step1_candidates": int(n_cand), "generate_s": round(t_gen, 2)})
            return best
    except Exception as e:                                           # the contract line is still printed
        print("bench.py: C4 one-shot leg failed: %r" % (e,), file=sys.stderr)
        return {"measured": "failed", "error": repr(e)}


# rocprofv3's names of the candidates for "dominant kernel"
PMC_KERNEL = {"k_tm_walk": "lsg::k_tm_walk", "k_tm_gather": "lsg::k_tm_gather", "k_tm_gather_count": "lsg::k_tm_gather_count", "k_tm_count_direct": "lsg::k_tm_count_direct",
              "k_tm_count_win": "lsg::k_tm_count_win"}
# kernels that are not part of a step (the generator, the memo table of the call stage: once per process)
NOT_IN_A_STEP = ("k_synth", "k_tail_table", "calib_")


def csrc_digest():
    """content hash of the kernel sources: profiles recorded for another build are not quoted as this build's traffic"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "longsom_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "longsom_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def recorded_traffic(kernel):
    """roofline.traffic: PMC counters cannot be read from inside this process, so the number comes from the newest committed
    rocprofv3 --pmc passes of this same command (profiles/rNN_pmc_traffic.json, tools/collect_profiles.sh) — but only when they
    were taken with THIS build of the kernels; otherwise traffic is null and traffic_stale says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None, None, None
    f = files[-1]
    try:
        d = json.load(open(f))
        k = next((v for n, v in d["kernels"].items() if n.split("#")[0].split("<")[0] == PMC_KERNEL[kernel]), None)
        rel = os.path.relpath(f, ROOT)
        if d.get("_csrc_sha1") != csrc_digest():
            return None, "%s was recorded for another build of longsom_amd/csrc (sha1 %s)" % (rel, str(d.get("_csrc_sha1"))[:12]), True, None
        step = None
        loads = d.get("_loads")
        if loads:                                                  # every kernel of a step, launches per step x bytes per launch
            per = {n.split("#")[0] + ("#" + n.split("#")[1] if n.startswith("rocprim") else ""): v["traffic_bytes"] * v["launches"] / loads
                   for n, v in d["kernels"].items() if v.get("traffic_bytes") and not any(x in n for x in NOT_IN_A_STEP)}
            step = {"kernels_GB": {n: round(b / 1e9, 3) for n, b in sorted(per.items(), key=lambda kv: -kv[1])}, "sum_bytes": sum(per.values())}
        if k and k.get("traffic_bytes"):
            return k["traffic_bytes"], rel + ": FETCH_SIZE (x calibrated gfx950 correction) + WRITE_SIZE, separate --pmc passes, bytes per launch", False, step
    except (OSError, ValueError, KeyError):
        pass
    return None, None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)          # (80 x 25.6 ms: the timed region keeps the GPU busy for 2 s)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=float, default=None, help="override the read count (development only; the reported config changes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="leave the C4 one-shot leg (config.c4_oneshot, after the timed steps, N=1 only) out")
    ap.add_argument("--no-recount", action="store_true", help="leave the re-counts of the resident store after the timed steps out (profiling runs: every kernel launch then belongs to a step)")
    ap.add_argument("--e2e-reads", type=float, default=None,
                    help="reads of the end-to-end leg (files in -> files out, after the timed steps, N=1 only); 0 = none; default 1e7 = the metric's whole BAM (none with --reads)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    global synth, CallParams, CountParams, Engine, region_shards, sub_model
    from longsom_amd import synth
    from longsom_amd._lib import CallParams, CountParams
    from longsom_amd.engine import Engine
    from longsom_amd.shard import region_shards, sub_model
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under torch.distributed.run with N ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a 1-GPU box: LSG_BENCH_DEVICE pins every rank to one device, LSG_BENCH_BACKEND=gloo moves the
    # collectives to the CPU (RCCL refuses two ranks on one GPU); the driver's runs use neither
    if "LSG_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["LSG_BENCH_DEVICE"])
    backend = os.environ.get("LSG_BENCH_BACKEND", "nccl")
    # LSG_BENCH_FORCE_DIST=1: bring the process group up and run the per-step all-gather even with ONE rank — the only way to put the
    # RCCL code path (device tensors, all_gather_into_tensor, barrier, all_reduce) through its paces on a one-GPU box
    dist_on = world > 1 or os.environ.get("LSG_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kw = {} if args.reads is None else {"n_reads": int(args.reads)}
    # the compact arrays as this package's own decoder (csrc/ingest.hip) leaves them in HBM: tile-phased (LSG_LAYOUT_PHASED,
    # include/longsom_hip.h) - a read's events inside one 64-position tile lie inside one aligned 128-byte line; LSG_BENCH_COMPACT=1:
    # segment after segment without gaps (round 4's input)
    phased = os.environ.get("LSG_BENCH_COMPACT") != "1"
    model = synth.named("C2", layout=1 if phased else 0, **kw)
    eng = Engine(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    eng.set_contigs(model.contig_len)
    eng.synth_reference(model.seed)
    eng.set_barcodes(model.celltype_of, 2)

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(eng, model)

    lo, hi, g_lo, g_hi = region_shards(model, world)[rank]
    # the rank's compact read-record arrays, generated once in HBM (untimed): what a device-resident decode of its share of the BAM hands over
    reads = eng.synth_generate(sub_model(model, g_lo, g_hi) if world > 1 else model)
    eng.set_region(lo[0], lo[1], hi[0], hi[1])
    n_reads, n_segs, n_events = int(reads.n_reads), int(reads.n_segs), int(reads.n_events)      # (n_events: the event array's extent, gaps included)

    class _SegLen:                                                 # the generated seg_len array where it lies, for one sum
        __cuda_array_interface__ = {"shape": (n_segs,), "typestr": "<i4", "data": (int(reads.seg_len), False), "version": 2}
    reads_events_sum = int(torch.as_tensor(_SegLen(), device=dev).sum(dtype=torch.int64).item()) if n_segs else 0
    cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
    # the load drops what SplitBam's MAPQ filter and the pileup's read filter drop (SplitBamCellTypes.py:110-113, BaseCellCounter.py:191,249):
    # in the product the host decode does (hostio.decode_bam(min_mapq=...)); the generated arrays hold every read of the BAM
    eng.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)
    # ... and, as in every fused rule, the count's parameters are known when the reads are loaded: the load makes the BAM's one count in the
    # pass that builds the store (lsg_set_count_at_load); LSG_BENCH_TWO_PASS=1 keeps the load and the count apart (round 3's step)
    # ... and a BAM is counted once: the load keeps no tile store (lsg_set_store_policy; the count reads the events where they lie).
    # LSG_BENCH_KEEP_STORE=1: the load also writes the store a later count would work on (k_tm_gather_count)
    keep_store = os.environ.get("LSG_BENCH_KEEP_STORE") == "1"
    if os.environ.get("LSG_BENCH_TWO_PASS") != "1":
        eng.set_count_at_load(cp)
        if not keep_store:
            eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
    # N > 1: ONE all-gather per step.  Every rank sends a message of the same agreed size: a header slot holding its number of
    # PASS-candidate rows, then room for cap_rows rows (SURVEY §8e's counts-then-buffers exchange needs two collectives and a host
    # read between them on every step).  The capacity is agreed during warm-up (the headers are read there) and checked once more
    # on the last timed step's buffer after the clock has stopped; a rank whose rows do not fit sends its count and no rows, so
    # every rank sees the overflow and the step is repeated with a larger capacity.
    gather = {"cap": int(os.environ.get("LSG_BENCH_GATHER_CAP", "64")), "send": None, "recv": None, "counts": None}

    def exchange(n_pass_hint=None):
        cap = gather["cap"]
        per = (cap + 1) * CALL_BYTES
        if gather["send"] is None or gather["send"].numel() != per:
            gather["send"] = torch.zeros(per, dtype=torch.uint8, device=dev)
            gather["recv"] = torch.zeros(world * per, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()                               # the engine's launches do not queue behind torch's fill
        send, recv = gather["send"], gather["recv"]
        try:
            n_pass = eng.export_calls(2, send.data_ptr() + CALL_BYTES, cap)
        except RuntimeError as e:
            if "capacity" not in str(e):
                raise
            n_pass = eng.export_calls(2)                         # count only: the header tells every rank to grow
        send[:8].view(torch.int64)[0] = n_pass
        if backend == "nccl":
            dist.all_gather_into_tensor(recv, send)
        else:
            out_cpu = torch.empty(world * per, dtype=torch.uint8)
            dist.all_gather_into_tensor(out_cpu, send.cpu())
            recv.copy_(out_cpu)
        return n_pass

    def gathered_counts():
        per = (gather["cap"] + 1) * CALL_BYTES
        return gather["recv"].view(world, per)[:, :8].contiguous().view(torch.int64).flatten().cpu().tolist()

    def step(check=False, load=True):
        if load:
            eng.load_reads_struct(reads)                               # a fresh tile store from the compact arrays
        rows, cols = eng.pileup_count(cp)
        n_sites, n_cand = eng.call_step1(kp)
        n_pass = 0
        if dist_on:
            n_pass = exchange()
            if check:
                counts = gathered_counts()
                while max(counts) > gather["cap"]:                 # same decision on every rank: the headers are identical everywhere
                    gather["cap"] = 2 * max(counts)
                    n_pass = exchange()
                    counts = gathered_counts()
                gather["counts"] = counts
        return rows, cols, n_sites, n_cand, n_pass

    for _ in range(max(args.warmup, 1)):                               # (at least one warm-up load: the first one allocates every buffer)
        step(check=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walk_ms, walk_bytes, path_bytes, gather_ms, gather_bytes = 0.0, 0.0, 0.0, 0.0, 0.0
    build_ms = np.zeros(4)
    fused, direct, lines, windows = False, False, False, False
    for _ in range(args.steps):
        rows, cols, n_sites, n_cand, n_pass = step()
        st = eng.count_stats()
        bt = eng.build_times()
        path = eng.layout_info()[0]
        fused, direct, lines, windows = path in (3, 4, 5, 6), path in (4, 5, 6), path in (5, 6), path == 6      # the load made the count (3: k_tm_gather_count, writing the store as well; 4: k_tm_count_direct, no store; 5: the same, every entry fetched as its one 128-byte line)
        walk_ms += st.ms_walk                                      # HIP events around the counting kernel: k_tm_gather_count, or k_tm_walk after a plain load
        # 2 B per admitted event + 24 B per admitted read + 168 B per emitted row - the counting kernel reads every event of
        # the counted region once and emits the rows of the single-job tiles; the plain gather (two-pass loads) reads every stored event once
        walk_bytes += 2.0 * st.n_events_admitted + 24.0 * st.n_reads_admitted + 168.0 * st.rows_by_kernel[1]
        path_bytes += 2.0 * st.n_events_admitted + 24.0 * st.n_reads_admitted + 168.0 * sum(rows)
        gather_ms += bt[3]
        gather_bytes += 2.0 * eng.store_shape()[2]
        build_ms += np.array(bt)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    _, layout_ms, layout_bytes = eng.layout_info()
    # after the clock: counts of the SAME resident store (the re-annotation loop's second pass, a parameter sweep)
    torch.cuda.synchronize()
    t_re = time.perf_counter()
    n_re = 0 if args.no_recount else 5
    re_walk_ms = 0.0
    store_build_ms = None
    if n_re and direct:                                             # the timed loads kept no store: one load that does, for the re-counts
        eng.set_store_policy(eng.STORE_KEEP)
        eng.load_reads_struct(reads)                                # (allocates the store: 24 GB at C2)
        eng.load_reads_struct(reads)                                # ... the same load, warm: what one more BAM costs this way
        eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
        store_build_ms = eng.layout_info()[1]
        t_re = time.perf_counter()
    for _ in range(n_re):
        re_rows, re_cols, re_sites, re_cand, _ = step(load=False)
        re_walk_ms += eng.count_stats().ms_walk
    torch.cuda.synchronize()
    recount_ms = (time.perf_counter() - t_re) / max(n_re, 1) * 1e3 if n_re else None
    if n_re:
        assert (re_rows, re_cols, re_sites, re_cand) == (rows, cols, n_sites, n_cand), "a re-count of the resident store differs from the first count"
    if os.environ.get("LSG_BENCH_STATS"):                           # the last step's counters, for whoever tunes the kernels
        print({f: (list(getattr(st, f)) if f.endswith("by_kernel") else getattr(st, f)) for f, _ in st._fields_ if f != "pad_"}, file=sys.stderr)
    shape_entries, _, shape_events = eng.store_shape()
    c4 = None
    if rank == 0 and world == 1 and args.reads is None and not args.no_c4:
        eng.unload_reads()
        for b in ("send", "recv"):
            gather[b] = None
        c4 = c4_oneshot(local_rank)
    e2e = None
    e2e_reads = int(args.e2e_reads) if args.e2e_reads is not None else (10_000_000 if args.reads is None else 0)
    if rank == 0 and world == 1 and e2e_reads > 0:
        e2e = end_to_end(e2e_reads)
    if dist_on:
        counts = gathered_counts()                                 # after the clock: the timed exchanges all fitted
        if max(counts) > gather["cap"]:
            raise RuntimeError("PASS-candidate rows outgrew the agreed all-gather capacity during the timed steps: %s > %d" % (counts, gather["cap"]))
        gather["counts"] = counts
    n_real_events = int(reads_events_sum)
    vals = torch.tensor([dt, float(cols), float(n_sites), float(n_cand), float(n_reads), float(n_events), float(sum(rows)), float(n_real_events)],
                        dtype=torch.float64, device=dev)
    if dist_on:
        if backend != "nccl":
            vals = vals.cpu()
        mx = vals.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item()); tot = sm.tolist()
    else:
        tot = vals.tolist()
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        sites = tot[1]
        count_kernel = "k_tm_count_win" if windows else "k_tm_count_direct" if direct else "k_tm_gather_count" if fused else "k_tm_walk"
        kernels = {count_kernel: {"avg_launch_ms": walk_ms / args.steps, "algorithmic_bytes_per_launch": walk_bytes / args.steps,
                                  "achieved_GBps": walk_bytes / max(walk_ms, 1e-9) / 1e6,
                                  "what": "SURVEY 8(d) bytes of the count it makes: 2 B x admitted events + 24 B x admitted reads + 168 B x rows it emits"}}
        if not fused:                                              # two-pass loads: the store build's gather is a kernel of its own (priced at the 2 B per stored event it reads)
            kernels["k_tm_gather"] = {"avg_launch_ms": gather_ms / args.steps, "algorithmic_bytes_per_launch": gather_bytes / args.steps,
                                      "achieved_GBps": gather_bytes / max(gather_ms, 1e-9) / 1e6}
        elif n_re:                                                 # the walk of the re-counts after the clock (same bytes, read from the store)
            kernels["k_tm_walk (re-count of the resident store, after the timed steps)"] = {
                "avg_launch_ms": re_walk_ms / n_re, "algorithmic_bytes_per_launch": walk_bytes / args.steps, "achieved_GBps": walk_bytes / args.steps / max(re_walk_ms / n_re, 1e-9) / 1e6}
        dom = max((k for k in kernels if "re-count" not in k), key=lambda k: kernels[k]["avg_launch_ms"])
        traffic, traffic_src, traffic_stale, step_traffic = (None, None, None, None)
        if world == 1 and args.reads is None:
            traffic, traffic_src, traffic_stale, step_traffic = recorded_traffic(dom)
        if step_traffic:
            step_traffic["over_algorithmic"] = step_traffic["sum_bytes"] / (path_bytes / args.steps)
        achieved = kernels[dom]["achieved_GBps"]
        bm = build_ms / args.steps
        out = {
            "metric": METRIC, "value": sites / (ms_step / 1e3), "unit": "sites/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2: whole-genome synthetic long-read workload (hg38/10 + chrM), %d reads x %d barcodes, 2 cell types; one step = "
                                   "one BAM's one-shot pass: device load of the compact read-record arrays (entries binned per tile and sorted by barcode%s) + pileup count + merge + step-1 call%s"
                                   % (model.n_reads, model.n_cb, "; no tile store kept" if direct else ", tile store written", " + RCCL all-gather of PASS-candidate call rows" if world > 1 else ""),
                       "reads": model.n_reads, "barcodes": model.n_cb, "reads_loaded_all_ranks": int(tot[4]), "events_loaded_all_ranks": int(tot[7]),
                       "input_layout": ("tile-phased (LSG_LAYOUT_PHASED): every segment at an event offset congruent to its reference start modulo 64, gaps of zeros; "
                                        "event array %.2f GB for %.2f GB of events" % (2 * tot[5] / 1e9, 2 * tot[7] / 1e9)) if phased else "compact: segment after segment",
                       "count_reads_whole_lines": bool(lines), "entries_binned_by_128_position_windows": bool(windows),
                       "sites_counted": int(sites), "rows_emitted": int(tot[6]), "merged_sites": int(tot[2]), "step1_candidates": int(tot[3]),
                       "sharding": "genomic regions balanced by estimated work" if world > 1 else "none",
                       "pass_rows_gathered": int(sum(gather["counts"])) if dist_on and gather["counts"] else None,
                       "exchange": "one all-gather per step (%s), %d-row slots agreed in warm-up" % (backend, gather["cap"]) if dist_on else None,
                       "path_algorithmic_GBps_rank0": path_bytes / dt / 1e9,
                       "step_parts_ms_rank0": {"load_wall": round(layout_ms, 2), "build_capacities_scatter": round(float(bm[0]), 2), "build_sort": round(float(bm[1]), 2),
                                               "build_block_tables": round(float(bm[2]), 2),
                                               ("build_plan_count" if direct else "build_plan_gather_count" if fused else "build_gather"): round(float(bm[3]), 2),
                                               ("count_kernel_inside_the_load" if fused else "count_walk"): round(float(st.ms_walk), 2),
                                               "count_total": round(float(st.ms_total), 2)},
                       "one_pass_load_and_count": bool(fused), "store_kept_by_the_timed_loads": not direct,
                       "load_that_also_writes_the_store_ms": None if store_build_ms is None else round(store_build_ms, 2),
                       "recount_ms": None if recount_ms is None else round(recount_ms, 2),                 # count + call over the SAME resident store: NOT what value is computed from
                       "resident_GB_rank0": round(layout_bytes / 1e9, 2),     # store + per-read / per-segment arrays + cached build temporaries
                       "store_entries_rank0": shape_entries, "store_events_rank0": shape_events,
                       "kernels": kernels, "c4_oneshot": c4, "end_to_end": e2e},                # measured in this run (or null): never a quoted file
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "avg_launch_ms": kernels[dom]["avg_launch_ms"], "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
                         # the whole step by SURVEY 8(d)'s bytes: (2 E + 24 R + 168 S_emit) / ms_per_step
                         "path_algorithmic_bytes_per_step": path_bytes / args.steps, "path_achieved": path_bytes / dt / 1e9, "path_frac": path_bytes / dt / 1e9 / HBM_PEAK_GBS,
                         "step_traffic": step_traffic},
        }
        if base is not None:
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
