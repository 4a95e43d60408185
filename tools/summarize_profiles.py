"""Condense gpurun_out/prof/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
  rNN_bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2`
  rNN_bench_under_rocprof.json    the bench line printed by that same run
  rNN_pmc_traffic.json            per kernel, per launch: FETCH_SIZE (raw KB and corrected bytes), WRITE_SIZE, the
                                  size-resolved TCC_EA0_RDREQ counters, and the FETCH_SIZE calibration
usage: python tools/summarize_profiles.py r01"""
import collections, csv, json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short_kernel_name

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/prof", "profiles"
os.makedirs(dst, exist_ok=True)
with open(f"{dst}/{tag}_bench_kernel_stats.csv", "w", newline="") as fo:      # rocprofv3 --stats, the kernels under short unique names (tools/kernel_names.py)
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in csv.DictReader(open(f"{src}/trace/bench_kernel_stats.csv")):
        w.writerow([short_kernel_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
line = [l for l in open(f"{src}/bench_under_rocprof.json") if l.startswith("{")][-1]
json.dump(json.loads(line), open(f"{dst}/{tag}_bench_under_rocprof.json", "w"), indent=1)


def counters(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


fetch, write, rdreq, calib = (counters(f"{src}/{n}/{f}_counter_collection.csv") for n, f in
                              (("fetch", "bench"), ("write", "bench"), ("rdreq", "bench"), ("calib", "calib")))
avg = lambda v: sum(v) / len(v) if v else None
cal = {}
for k, v in calib.items():
    if k.startswith("calib_"):
        cal[k.split("(")[0]] = {"bytes_read_per_launch": 4 << 30, "FETCH_SIZE_KB_avg": avg(v["FETCH_SIZE"]),
                                "true_over_reported": (4 << 30) / (avg(v["FETCH_SIZE"]) * 1024)}
factor = cal.get("calib_ushort", {}).get("true_over_reported", 2.0)
out = {"_method": "separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ by size) of `python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline`; per-launch averages.  FETCH_SIZE under-reports 128-byte requests on gfx950 (MI355X_MICROARCH.md, HBM): "
                  "corrected by the factor measured with tools/pmc_calib.hip on the walk kernel's own access pattern (2-byte-per-lane buffer "
                  "loads) and cross-checked against 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B.  WRITE_SIZE is taken as reported.",
       "_calibration": cal, "_fetch_correction": factor, "kernels": {}}
# which build of the kernels this was measured on: bench.py quotes the traffic only for the same sources (bench.csrc_digest)
sys.path.insert(0, ".")
import bench
out["_csrc_sha1"] = bench.csrc_digest()
# loads (= steps incl. warm-up) of the profiled run: launches / _loads = a kernel's launches per step (the run makes no re-counts: --no-recount)
out["_loads"] = max((len(v["FETCH_SIZE"]) for k, v in fetch.items() if "k_seg_static" in k), default=None)
for k in sorted(set(fetch) | set(write) | set(rdreq)):
    f, w = avg(fetch[k]["FETCH_SIZE"]), avg(write[k]["WRITE_SIZE"])
    r = rdreq.get(k, {})
    sized = None
    if r:
        sized = 32 * avg(r["TCC_EA0_RDREQ_32B_sum"]) + 64 * avg(r["TCC_EA0_RDREQ_64B_sum"]) + 128 * avg(r["TCC_EA0_RDREQ_128B_sum"])
    if (f or 0) * 1024 < 2e6 and (w or 0) * 1024 < 2e6:
        continue
    out["kernels"][short_kernel_name(k)] = {
        "launches": len(fetch[k]["FETCH_SIZE"]), "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
        "read_bytes_corrected": None if f is None else f * 1024 * factor, "read_bytes_from_sized_rdreq": sized,
        "write_bytes": None if w is None else w * 1024,
        "traffic_bytes": None if f is None or w is None else f * 1024 * factor + w * 1024}
json.dump(out, open(f"{dst}/{tag}_pmc_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-34s read %.2f GB (sized %.2f)  write %.2f GB" % (k[:34], (v["read_bytes_corrected"] or 0) / 1e9, (v["read_bytes_from_sized_rdreq"] or 0) / 1e9, (v["write_bytes"] or 0) / 1e9))
