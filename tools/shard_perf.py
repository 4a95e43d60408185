"""Per-rank time of ONE STEP of the sharded C2 job as bench.py --gpus N times it - the rank's load (which makes its count in the same
pass and keeps no store: lsg_set_count_at_load + lsg_set_store_policy), the count's hand-over, merge + step-1 call, export of its PASS-candidate rows into the send buffer -
emulated on ONE GPU: one shard after another, no collective.  max over the ranks of a given N predicts bench.py --gpus N minus the
all-gather (a few hundred rows over xGMI) and whatever eight processes sharing one host cost.  Writes gpurun_out/shard_perf.json.
usage: python tools/shard_perf.py [n_reads] [worlds, e.g. 1,2,4,8]"""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from longsom_amd import synth
from longsom_amd._lib import CallParams, CountParams
from longsom_amd.engine import Engine
from longsom_amd.shard import region_shards, sub_model, CALL_BYTES

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
model = synth.named("C2", n_reads=n_reads, layout=0 if os.environ.get("LSG_BENCH_COMPACT") == "1" else 1)      # (as bench.py generates it: tile-phased events)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
cp, kp = CountParams.longsom_defaults(), CallParams.longsom_defaults()
eng.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)
eng.set_count_at_load(cp)
eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)      # (bench.py's step: the BAM is counted once, no store kept)
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
# ONE measured RCCL all-gather of bench.py's message (a header slot + 64 rows of CALL_BYTES) in a one-rank group: what the collective's own
# launch and completion cost a rank's step (the transfer itself - a few KB over xGMI - is not what a one-rank group can show)
coll_ms = None
try:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1)
    send = torch.zeros(65 * CALL_BYTES, dtype=torch.uint8, device="cuda"); recv = torch.zeros_like(send)
    for _ in range(5):
        dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize(); coll_ms = (time.perf_counter() - t0) / 50 * 1e3
    dist.destroy_process_group()
    print("one-rank RCCL all_gather_into_tensor of %d bytes: %.3f ms" % (send.numel(), coll_ms), flush=True)
except Exception as e:      # noqa: BLE001
    print("RCCL one-rank group failed:", repr(e), flush=True)
out = {"workload": "C2 at %d reads, one step = load (+ count in the same pass) + call + export per rank, ranks emulated one after another on one GPU" % n_reads, "worlds": {}}
for world in worlds:
    per_rank = []
    for rank, (lo, hi, g_lo, g_hi) in enumerate(region_shards(model, world)):
        reads = eng.synth_generate(sub_model(model, g_lo, g_hi) if world > 1 else model)      # the rank's compact arrays, resident (untimed)
        eng.set_region(lo[0], lo[1], hi[0], hi[1])

        def step():
            eng.load_reads_struct(reads)
            rows, cols = eng.pileup_count(cp)
            ns, nc = eng.call_step1(kp)
            npass = eng.export_calls(2, buf.data_ptr(), buf.numel() // CALL_BYTES)      # as bench.py does: rows straight into the send buffer
            return cols, ns, nc, npass
        step(); step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cols, ns, nc, npass = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        st = eng.count_stats(); bt = eng.build_times()
        per_rank.append({"rank": rank, "ms_per_step": round(ms, 3), "reads": int(reads.n_reads), "columns": int(cols), "merged_sites": int(ns), "pass_rows": int(npass),
                         "load_kernels_ms": [round(float(x), 3) for x in bt], "count_kernel_ms": round(float(st.ms_walk), 3), "load_path": eng.layout_info()[0]})
        print(f"N={world} rank={rank}: {ms:.2f} ms  reads {reads.n_reads} cols {cols} sites {ns} cand {nc} pass {npass} | build {[round(float(x), 2) for x in bt]} count kernel {st.ms_walk:.2f}", flush=True)
    worst = max(r["ms_per_step"] for r in per_rank)
    out["worlds"][str(world)] = {"slowest_rank_ms": worst, "ranks": per_rank}
    print(f"N={world}: slowest rank {worst:.2f} ms", flush=True)
w1 = out["worlds"].get("1", {}).get("slowest_rank_ms")
out["rccl_all_gather_one_rank_ms"] = None if coll_ms is None else round(coll_ms, 4)
if w1:
    out["predicted_speedup_vs_1"] = {k: round(w1 / v["slowest_rank_ms"], 2) for k, v in out["worlds"].items()}
    if coll_ms is not None:
        out["predicted_speedup_vs_1_with_the_collective"] = {k: round(w1 / (v["slowest_rank_ms"] + (coll_ms if k != "1" else 0.0)), 2) for k, v in out["worlds"].items()}
    print("predicted speed-up (slowest rank, no collective):", out["predicted_speedup_vs_1"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shard_perf.json", "w"), indent=1)
