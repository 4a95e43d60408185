#!/bin/bash
# Run on the GPU box (gpurun): everything profiles/rNN_* is made of, in one call.  Results under gpurun_out/final/ (copied into profiles/ by hand).
#   usage: bash tools/final_profiles.sh r03
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final; rm -rf $F; mkdir -p $F
echo "== rocprof passes"; bash tools/collect_profiles.sh > $F/collect.log 2>&1
mkdir -p $F/p; python3 - "$TAG" <<'PY'
import subprocess, sys, shutil, glob, os
tag = sys.argv[1]
subprocess.check_call([sys.executable, "tools/summarize_profiles.py", tag])
for f in glob.glob("profiles/%s_bench_kernel_stats.csv" % tag) + glob.glob("profiles/%s_bench_under_rocprof.json" % tag) + glob.glob("profiles/%s_pmc_traffic.json" % tag):
    shutil.copy(f, "gpurun_out/final/")
PY
# (after the counter passes: bench.py quotes roofline.traffic from the profiles/${TAG}_pmc_traffic.json the lines above just wrote for this build)
echo "== default bench"; timeout -k 10 500 python3 bench.py > $F/${TAG}_bench_default_run.json 2> $F/bench_default.err; tail -c 600 $F/${TAG}_bench_default_run.json; echo
echo "== the store-keeping step (k_tm_gather_count) and the re-counts over its store (k_tm_resolve, k_tm_walk)"
export LSG_BENCH_KEEP_STORE=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/keep -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-c4 > $F/${TAG}_keep_store_bench.json 2> $F/keep.err || echo "keep-store trace failed"
unset LSG_BENCH_KEEP_STORE
python3 - "$TAG" <<'PY'
import csv, sys
sys.path.insert(0, "tools")
from kernel_names import short_kernel_name
tag = sys.argv[1]
with open("gpurun_out/final/%s_keep_store_kernel_stats.csv" % tag, "w", newline="") as fo:
    w = csv.writer(fo); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in csv.DictReader(open("gpurun_out/prof/keep/bench_kernel_stats.csv")):
        w.writerow([short_kernel_name(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
echo "== timeline"; python3 tools/timeline.py gpurun_out/prof/trace > $F/${TAG}_timeline.txt; tail -3 $F/${TAG}_timeline.txt
echo "== sq counters"; bash tools/sq_counters.sh > $F/${TAG}_sq_counters.txt 2> $F/sq.err; wc -l $F/${TAG}_sq_counters.txt
echo "== shard_perf"; timeout -k 10 400 python3 tools/shard_perf.py > $F/shard_perf.log 2>&1; cp gpurun_out/shard_perf.json $F/${TAG}_shard_perf.json; tail -2 $F/shard_perf.log
echo "== done"
