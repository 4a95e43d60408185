#!/bin/bash
# Run on the GPU box: one rocprofv3 --pmc pass (read requests to the fabric by size) of a short bench run; prints the per-launch bytes of the kernels named on the command line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/qt; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4 > $O/out 2> $O/err || { echo failed; tail -3 $O/err; exit 1; }
python3 - "$@" <<'PY'
import csv, glob, collections, sys
want = sys.argv[1:] or ["count_direct"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/qt/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, "read GB per launch: %.2f" % ((32 * a.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * a.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * a.get("TCC_EA0_RDREQ_128B_sum", 0)) / 1e9), {c: int(x) for c, x in a.items()})
PY
