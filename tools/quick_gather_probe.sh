cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LSG_TIMING=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b4.log 2> gpurun_out/b4.err; echo rc=$?
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/b4.log') if l.startswith('{')][-1])
print(d['ms_per_step'], d['config']['step_parts_ms_rank0'], d['config']['recount_ms'], d['roofline']['kernel'], d['roofline']['frac'])
"
tail -1 gpurun_out/b4.err
O=gpurun_out/prof_q; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rdreq.out 2> $O/rdreq.err
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_q/rdreq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "tm_gather" in k or "tm_walk" in k:
        a = lambda n: sum(v[n]) / len(v[n])
        print(k, "read GB", (32 * a("TCC_EA0_RDREQ_32B_sum") + 64 * a("TCC_EA0_RDREQ_64B_sum") + 128 * a("TCC_EA0_RDREQ_128B_sum")) / 1e9)
PY
