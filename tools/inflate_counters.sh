#!/bin/bash
# Run on the GPU box: SQ counters of the device inflate (k_inflate) over a 1 M-read BAM's ingest, one rocprofv3 --pmc pass per set.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sqi; rm -rf $O; mkdir -p $O
B="python3 tools/e2e_perf.py 1e6 0"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); echo "set $i: $set"
  timeout -k 10 250 rocprofv3 --pmc $set --output-format csv -d $O/s$i -o run -- $B > $O/s$i.out 2> $O/s$i.err || echo "set $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/sqi/s*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "k_inflate" in k or "k_rec_emit" in k or "k_block_crc" in k:
                print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
