#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its named configuration.

  metric   genomic sites/s, pileup + call: pileup columns with >= 1 counted entry, summed over cell
           types, per second of one full pass (binning -> pileup count -> merge + step-1 call,
           + the all-gather of PASS-candidate call tables when N > 1)
  workload C2 (BASELINE.json configs[1]): whole-genome synthetic long-read workload, 10 M reads x 5 k
           barcodes, generated directly in HBM by the model of longsom_amd/csrc/synth_model.h
           (inputs are resident when the timed region starts)
  N > 1    strong scaling: the same 10 M-read workload, genomic windows sharded over the ranks by
           read count; every rank loads the reads overlapping its region, counts only its own
           columns and the ranks all-gather their PASS-candidate call rows over RCCL.

One JSON line on rank 0.  `roofline` is for the dominant kernel (k_walk_block): algorithmic bytes
(2 B/event + 24 B/read + 168 B/emitted row, restricted to what that kernel processes)
over its HIP-event time on its own stream.  `cpu_baseline` times the CPU oracle (oracle/, kind
"port") on a bounded sample of the same workload, single thread, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from longsom_amd import synth  # noqa: E402
from longsom_amd._lib import CallParams, CountParams  # noqa: E402
from longsom_amd.engine import Engine  # noqa: E402

METRIC = "genomic sites/s pileup+call, 10M-read BAM x 5k barcodes, 1/2/4/8 MI355X"
CALL_BYTES = 336          # sizeof(lsg_call)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


from longsom_amd.shard import region_shards, sub_model  # noqa: E402


def cpu_baseline(eng, model, target_reads=150_000, call_sites=10_000):
    """Oracle (C count + Python/scipy step 1, single thread) on a contiguous-gene sample of the workload.
    Also checks the GPU result on that sample against the oracle (a parity check at bench time)."""
    from oracle import calling_oracle, loader
    from longsom_amd import tsvio
    reads = np.diff(model.gene_read_off)
    g_lo = model.n_genes // 3
    g_hi = g_lo
    while g_hi < model.n_genes and int(reads[g_lo:g_hi].sum()) < target_reads:
        g_hi += 1
    sm = sub_model(model, g_lo, g_hi)
    eng.set_region()
    eng.synth_reads(sm)
    rows, cols = eng.pileup_count()
    rec = eng.reads_to_host()
    tids = sorted(set(sm.gene_tid.tolist()))
    refs = [eng.reference_to_host(t) if t in tids else np.zeros(0, np.uint8) for t in range(len(model.contig_len))]
    t0 = time.time()
    ok, per_ct, n_cols = True, [], 0
    for ct in range(2):
        k, rf, c, ncol = loader.count(rec, model.contig_len, refs, model.celltype_of, ct)
        per_ct.append((k, rf, c)); n_cols += ncol
    t_count = time.time() - t0
    for ct in range(2):
        gk, gr, gc = eng.fetch_counts(ct)
        ok &= bool(np.array_equal(gk, per_ct[ct][0]) and np.array_equal(gc, per_ct[ct][2]))
    ok &= n_cols == cols
    # step 1 on a bounded number of merged sites (scipy betabinom per alt, as the reference does)
    sub = [(k[:call_sites], r[:call_sites], c[:call_sites]) for k, r, c in per_ct]
    merged = tsvio.format_merged_tsv(sub, model.contig_names, ["Cancer", "Non-Cancer"])
    n_call = sum(1 for l in merged.split("\n") if l and not l.startswith("#"))
    fasta = {model.contig_names[t]: refs[t].tobytes().decode() for t in tids}
    t0 = time.time()
    calling_oracle.step1(merged, fasta, info_lines=tsvio.STEP1_INFO_LINES)
    t_call_site = (time.time() - t0) / max(1, n_call)
    n_merged = len(np.unique(np.concatenate([p[0] for p in per_ct]))) if n_cols else 0
    t_total = t_count + t_call_site * n_merged
    return {"value": n_cols / t_total if t_total > 0 else 0.0, "unit": "sites/s", "cores": 1, "kind": "port",
            "sample": "%d reads of %d contiguous genes of the C2 workload (%d events, %d columns): oracle/count_oracle.c timed on all of "
                      "it (%.1f s), oracle/calling_oracle.py step1 timed on %d merged sites (%.2f ms/site) and scaled to the sample's %d sites"
                      % (rec.n_reads, g_hi - g_lo, rec.n_events, n_cols, t_count, n_call, t_call_site * 1e3, n_merged),
            "gpu_matches_oracle_on_sample": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=float, default=None, help="override the read count (development only; the reported config changes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a 1-GPU box: LSG_BENCH_DEVICE pins every rank to one device, LSG_BENCH_BACKEND=gloo moves the
    # collectives to the CPU (RCCL refuses two ranks on one GPU); the driver's runs use neither
    if "LSG_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["LSG_BENCH_DEVICE"])
    backend = os.environ.get("LSG_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kw = {} if args.reads is None else {"n_reads": int(args.reads)}
    model = synth.named("C2", **kw)
    eng = Engine(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    eng.set_contigs(model.contig_len)
    eng.synth_reference(model.seed)
    eng.set_barcodes(model.celltype_of, 2)

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(eng, model)

    lo, hi, g_lo, g_hi = region_shards(model, world)[rank]
    eng.synth_reads(sub_model(model, g_lo, g_hi) if world > 1 else model)
    eng.set_region(lo[0], lo[1], hi[0], hi[1])
    n_reads, n_segs, n_events = eng.reads_shape()
    cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
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

    def step(check=False):
        rows, cols = eng.pileup_count(cp)
        n_sites, n_cand = eng.call_step1(kp)
        n_pass = 0
        if world > 1:
            n_pass = exchange()
            if check:
                counts = gathered_counts()
                while max(counts) > gather["cap"]:                 # same decision on every rank: the headers are identical everywhere
                    gather["cap"] = 2 * max(counts)
                    n_pass = exchange()
                    counts = gathered_counts()
                gather["counts"] = counts
        return rows, cols, n_sites, n_cand, n_pass

    for _ in range(max(args.warmup, 1) if world > 1 else args.warmup):
        step(check=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walk_ms, walk_bytes, path_bytes = 0.0, 0.0, 0.0
    for _ in range(args.steps):
        rows, cols, n_sites, n_cand, n_pass = step()
        st = eng.count_stats()
        e_walk, r_walk = st.events_by_kernel[1], st.rows_by_kernel[1]
        e_tot = max(1, st.n_events_admitted)
        walk_ms += st.ms_walk
        walk_bytes += 2.0 * e_walk + 24.0 * st.n_reads_admitted * (e_walk / e_tot) + 168.0 * r_walk
        path_bytes += 2.0 * st.n_events_admitted + 24.0 * st.n_reads_admitted + 168.0 * sum(rows)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        counts = gathered_counts()                                 # after the clock: the timed exchanges all fitted
        if max(counts) > gather["cap"]:
            raise RuntimeError("PASS-candidate rows outgrew the agreed all-gather capacity during the timed steps: %s > %d" % (counts, gather["cap"]))
        gather["counts"] = counts
    vals = torch.tensor([dt, float(cols), float(n_sites), float(n_cand), float(n_reads), float(n_events), float(sum(rows))],
                        dtype=torch.float64, device=dev)
    if world > 1:
        if backend != "nccl":
            vals = vals.cpu()
        mx = vals.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0].item()); tot = sm.tolist()
    else:
        tot = vals.tolist()
    if rank == 0:
        # HBM-side bytes of the dominant kernel per launch: PMC counters cannot be read from inside this process, so the
        # number comes from the committed rocprofv3 --pmc passes of this same command (profiles/, tools/collect_profiles.sh)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if world == 1 and args.reads is None and os.path.exists(pmc):
            try:
                k = json.load(open(pmc))["kernels"].get("lsg::k_walk_block")
                if k and k.get("traffic_bytes"):
                    traffic = k["traffic_bytes"]
                    traffic_src = "profiles/r01_pmc_traffic.json: FETCH_SIZE (x calibrated gfx950 correction) + WRITE_SIZE, separate --pmc passes, bytes per launch"
            except (OSError, ValueError, KeyError):
                pass
        ms_step = dt / args.steps * 1e3
        sites = tot[1]
        achieved = walk_bytes / max(walk_ms, 1e-9) / 1e6         # GB/s
        out = {
            "metric": METRIC, "value": sites / (ms_step / 1e3), "unit": "sites/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2: whole-genome synthetic long-read workload (hg38/10 + chrM), %d reads x %d barcodes, 2 cell types, "
                                   "pileup count + merge + step-1 call%s" % (model.n_reads, model.n_cb, ", RCCL all-gather of PASS-candidate call rows" if world > 1 else ""),
                       "reads": model.n_reads, "barcodes": model.n_cb, "reads_loaded_all_ranks": int(tot[4]), "event_slots_resident_all_ranks": int(tot[5]),
                       "sites_counted": int(sites), "rows_emitted": int(tot[6]), "merged_sites": int(tot[2]), "step1_candidates": int(tot[3]),
                       "sharding": "genomic regions balanced by read count" if world > 1 else "none",
                       "pass_rows_gathered": int(sum(gather["counts"])) if world > 1 and gather["counts"] else None,
                       "exchange": "one all-gather per step, %d-row slots agreed in warm-up" % gather["cap"] if world > 1 else None,
                       "path_algorithmic_GBps_rank0": path_bytes / dt / 1e9},
            "roofline": {"bound": "hbm", "kernel": "k_walk_block", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": walk_ms / args.steps, "algorithmic_bytes_per_launch": walk_bytes / args.steps},
        }
        if base is not None:
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
