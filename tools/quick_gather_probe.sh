# scratch: step time + fabric traffic and L2 hit counters of k_tm_gather / k_tm_walk, once per environment setting given (usage: gpurun -- 'bash tools/quick_gather_probe.sh "A=0" "A=1"'; what the gather experiments of DESIGN.md §9 were measured with)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  export $V
  echo "== $V"
  LSG_TIMING=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --e2e-reads 0 --no-c4 > gpurun_out/b4.log 2> gpurun_out/b4.err || { echo bench failed; tail -5 gpurun_out/b4.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/b4.log') if l.startswith('{')][-1])
print(d['ms_per_step'], d['config']['step_parts_ms_rank0'], d['config']['recount_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['kernels'])
"
  O=gpurun_out/prof_q; rm -rf $O; mkdir -p $O
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-c4 > $O/rdreq.out 2> $O/rdreq.err || { echo prof failed; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/hit -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-c4 > $O/hit.out 2> $O/hit.err || { echo prof2 failed; exit 1; }
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_q/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "tm_gather" in k or "tm_walk" in k:
        a = lambda n: (sum(v[n]) / len(v[n])) if v.get(n) else float("nan")
        print(k[:40], "read GB %.2f" % ((32 * a("TCC_EA0_RDREQ_32B_sum") + 64 * a("TCC_EA0_RDREQ_64B_sum") + 128 * a("TCC_EA0_RDREQ_128B_sum")) / 1e9),
              "req32 %.3g req64 %.3g req128 %.3g all %.3g" % (a("TCC_EA0_RDREQ_32B_sum"), a("TCC_EA0_RDREQ_64B_sum"), a("TCC_EA0_RDREQ_128B_sum"), a("TCC_EA0_RDREQ_sum")),
              "hit %.3g miss %.3g req %.3g read %.3g" % (a("TCC_HIT_sum"), a("TCC_MISS_sum"), a("TCC_REQ_sum"), a("TCC_READ_sum")))
PY
done
