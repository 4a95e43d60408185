"""Configuration C4 of BASELINE.json (50 M reads x 20 k barcodes) on one GPU: does it fit, how long does a pass take, and do two
size-independent properties hold (region split invariance of the totals; the call stage covers every counted site)."""
import sys, time
sys.path.insert(0, ".")
import torch
from longsom_amd import synth
from longsom_amd.engine import Engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
model = synth.named("C4", n_reads=n)
eng = Engine(0, stream=torch.cuda.current_stream().cuda_stream)
eng.set_contigs(model.contig_len); eng.synth_reference(model.seed); eng.set_barcodes(model.celltype_of, 2)
from longsom_amd._lib import CountParams
cp = CountParams.longsom_defaults()
eng.set_load_filter(cp.min_mq, cp.flag_exclude, cp.ignore_orphans)          # as the product's decode and bench.py do
if len(sys.argv) > 2 and sys.argv[2] == "direct":
    # ONE-SHOT steps as bench.py times C2: the load makes the BAM's one count in its own pass and keeps no store (C4's 64-position tiles hold
    # more than max_depth = 200 000 reads, but no position does: lsg_max_live_reads_exact lets the load count).  Step 0 allocates and makes the
    # tail table of the parameter set (what bench.py's warm-up pays); the later ones are the figure.
    from longsom_amd._lib import CallParams
    kp = CallParams.longsom_defaults()
    eng.set_region()
    eng.set_count_at_load(cp); eng.set_store_policy(eng.STORE_SKIP_WHEN_COUNTED)
    for i in range(3):
        t0 = time.perf_counter(); eng.synth_reads(model); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        load_ms = eng.layout_info()[1]
        t0 = time.perf_counter(); rows, cols = eng.pileup_count(cp); ns, nc = eng.call_step1(kp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = eng.count_stats()
        print("one-shot C4 step %d (load path %d): lsg_load_reads %.1f ms (build phases %s, count kernel %.1f) + count hand-over and call %.1f ms = %.1f ms; %.3e sites/s; rows %s cols %d sites %d cand %d  [generate + load %.2f s wall]"
              % (i, eng.layout_info()[0], load_ms, [round(x, 2) for x in eng.build_times()], st.ms_walk, dt * 1e3, load_ms + dt * 1e3, cols / ((load_ms + dt * 1e3) / 1e3), rows, cols, ns, nc, dt2), flush=True)
    print("max live reads: tiles, all reads %d; tiles, per cell type %d; positions, per cell type %d; resident %.1f GB" % (eng.max_live_reads_all(), eng.max_live_reads(), eng.max_live_reads_exact(), eng.layout_info()[2] / 1e9), flush=True)
    sys.exit(0)
t0 = time.perf_counter(); eng.synth_reads(model); torch.cuda.synchronize(); dt = time.perf_counter() - t0      # generate + load + give the generated arrays back
n_reads, n_segs, n_events = eng.reads_shape()
print("generated and loaded in %.2f s: %d reads %d segments %d events (%.1f GB of compact events, not kept)" % (dt, n_reads, n_segs, n_events, n_events * 2 / 1e9), flush=True)
print("load: %.1f ms wall in lsg_load_reads (first load of the process: allocates the store); build kernels (capacities + scatter, sort, entry words, gather) %s ms; "
      "store %s (entries, blocks, events); resident %.1f GB; free %.0f GB" %
      (eng.layout_info()[1], [round(x, 2) for x in eng.build_times()], eng.store_shape(), eng.layout_info()[2] / 1e9, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
# the same load again, warm (every buffer of the store and of the build is there): what one more C4 BAM costs this process
t0 = time.perf_counter(); eng.synth_reads(model); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
load_warm_ms = eng.layout_info()[1]
print("second load (warm): generate + load %.2f s; lsg_load_reads %.1f ms wall, build kernels %s ms" % (dt2, load_warm_ms, [round(x, 2) for x in eng.build_times()]), flush=True)
# the first count + call of the process pays what bench.py's warm-up pays: row and call buffers, the tail table of the parameter set
t0 = time.perf_counter(); rows, cols = eng.pileup_count(); ns, nc = eng.call_step1(); torch.cuda.synchronize()
print("first count + call of the process (allocations, k_tail_table: what a warm-up pays): %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for i in range(3):
    t0 = time.perf_counter(); rows, cols = eng.pileup_count(); ns, nc = eng.call_step1(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = eng.count_stats()
    print("pass %d (count + call over the resident store): %.1f ms  rows %s cols %d sites %d cand %d | events %d entries %d units %d multi-job %d | resolve %.1f walk %.1f count %.1f ms | resident %.1f GB" %
          (i, dt * 1e3, rows, cols, ns, nc, s.n_events_admitted, s.n_entries, s.n_units, s.n_deep_units, s.ms_bin, s.ms_walk, s.ms_total, eng.layout_info()[2] / 1e9), flush=True)
print("one-shot C4 step through the store, warm = lsg_load_reads %.1f ms + count and call %.1f ms = %.1f ms (parts timed apart: a third load does not fit beside the rows; `tools/c4_run.py 5e7 direct` times the one-shot steps that keep no store)" % (load_warm_ms, dt * 1e3, load_warm_ms + dt * 1e3), flush=True)
full = (rows, cols, ns)
# property: counting two halves of the genome separately gives the same totals
tid_mid = len(model.contig_len) // 2
tot_rows, tot_cols, tot_sites = [0, 0], 0, 0
for lo, hi in (((0, 0), (tid_mid, 0)), ((tid_mid, 0), (len(model.contig_len), 0))):
    eng.set_region(lo[0], lo[1], hi[0], hi[1]); r, c = eng.pileup_count(); s_, _ = eng.call_step1()
    tot_rows = [a + b for a, b in zip(tot_rows, r)]; tot_cols += c; tot_sites += s_
print("split totals", tot_rows, tot_cols, tot_sites, "match" if (tot_rows == list(full[0])[:2] and tot_cols == full[1] and tot_sites == full[2]) else "MISMATCH", flush=True)

# ---- C4 as specified: the position sets resident in HBM beside the reads, every step-1 candidate probed (GetExtraFilters' membership
# tests, BaseCellCalling.step2.py:142-158).  Queries and hits stay on the device; the probe is timed with CUDA events on the engine's stream.
import json
import numpy as np
from tests.support import possets
eng.set_region()
rows, cols = eng.pileup_count(); ns, nc = eng.call_step1()
n_q = eng.export_calls(1)
from longsom_amd import _lib
buf = torch.zeros(n_q * 336, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()          # the engine runs on its OWN stream (a NULL stream handle selects it): the fill above must have landed
assert eng.export_calls(1, buf.data_ptr(), n_q) == n_q
keys = buf.view(n_q, 336)[:, :8].contiguous().view(torch.int64).flatten() + 1          # lsg_call.key is the record's first field
q_host = keys.cpu().numpy()
res = {"reads": n, "step2_rows_probed": int(n_q), "sets": {}}
hits = torch.zeros(n_q, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for kind, (name, seed, size, frac) in enumerate((("editing", 40, possets.C4_SIZES["editing"], 0.01), ("pon_SR", 41, possets.C4_SIZES["pon"], 0.03))):
    ks = possets.random_keys(seed, size, model.contig_len, salt=q_host, salt_frac=frac)
    t0 = time.time(); eng.load_posset(kind, ks); t_load = time.time() - t0
    best = 1e9
    for rep in range(3):              # lsg_probe_posset returns after its stream has drained: host clock around the call
        t0 = time.perf_counter(); eng.probe_posset_device(kind, keys.data_ptr(), n_q, hits.data_ptr()); best = min(best, (time.perf_counter() - t0) * 1e3)
    want = np.zeros(n_q, np.uint8); at = np.searchsorted(ks, q_host); ok = at < len(ks); want[ok] = ks[at[ok]] == q_host[ok]
    got = hits.cpu().numpy()
    res["sets"][name] = {"keys": int(len(ks)), "load_s": round(t_load, 3), "probe_ms": round(best, 3), "hits": int(got.sum()), "equals_numpy": bool(np.array_equal(got, want)),
                         "queries_per_s": n_q / (best / 1e3), "GBps_keys_touched": n_q * np.log2(max(2, len(ks))) * 8 / (best / 1e3) / 1e9}
    print(name, res["sets"][name], flush=True)
print("free HBM with reads + sets resident: %.0f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/c4_possets.json", "w"), indent=1)
