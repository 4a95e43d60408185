#!/bin/bash
# Run on the GPU box (gpurun): everything profiles/rNN_* is made of, in one call.  Results under gpurun_out/final/ (copied into profiles/ by hand).
#   usage: bash tools/final_profiles.sh r03
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final; rm -rf $F; mkdir -p $F
echo "== default bench"; timeout -k 10 500 python3 bench.py > $F/${TAG}_bench_default_run.json 2> $F/bench_default.err; tail -c 600 $F/${TAG}_bench_default_run.json; echo
echo "== rocprof passes"; bash tools/collect_profiles.sh > $F/collect.log 2>&1
mkdir -p $F/p; python3 - "$TAG" <<'PY'
import subprocess, sys, shutil, glob, os
tag = sys.argv[1]
subprocess.check_call([sys.executable, "tools/summarize_profiles.py", tag])
for f in glob.glob("profiles/%s_bench_kernel_stats.csv" % tag) + glob.glob("profiles/%s_bench_under_rocprof.json" % tag) + glob.glob("profiles/%s_pmc_traffic.json" % tag):
    shutil.copy(f, "gpurun_out/final/")
PY
echo "== timeline"; python3 tools/timeline.py gpurun_out/prof/trace > $F/${TAG}_timeline.txt; tail -3 $F/${TAG}_timeline.txt
echo "== sq counters"; bash tools/sq_counters.sh > $F/${TAG}_sq_counters.txt 2> $F/sq.err; wc -l $F/${TAG}_sq_counters.txt
echo "== done"
