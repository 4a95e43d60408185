#!/bin/bash
# Run on the GPU box: instruction-mix counters of the count kernels (one rocprofv3 --pmc pass per set).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE"; do
  i=$((i+1)); echo "set $i: $set"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/s$i -o bench -- $B > $O/s$i.out 2> $O/s$i.err || echo "set $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/sq/s*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "tm_walk" in k or "tm_resolve" in k or "tm_gather" in k or "tm_count" in k or "tm_fill" in k or "k_bin" in k or "call_gather" in k or "k_seg_static" in k or "segmented" in k[:400]:
                print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
