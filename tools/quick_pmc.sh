#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass of a short bench run with the counters named in $PMC; prints the per-launch averages of the kernels named on the command line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/qp; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d $O -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4 > $O/out 2> $O/err || { echo failed; tail -5 $O/err; exit 1; }
python3 - "$@" <<'PY'
import csv, glob, collections, sys
want = sys.argv[1:] or ["count_direct"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/qp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: int(sum(x) / len(x)) for c, x in v.items()})
PY
