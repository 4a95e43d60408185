#!/bin/bash
# Run on the GPU box (gpurun): a short bench per environment setting; prints ms_per_step and the step's parts.  usage: bash tools/sweep_env.sh "A=0" "A=1 B=2" ...
cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  echo "== $V"
  env $V LSG_TIMING=1 timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --e2e-reads 0 --no-c4 > gpurun_out/sw.log 2> gpurun_out/sw.err || { echo bench failed; tail -5 gpurun_out/sw.err; continue; }
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/sw.log') if l.startswith('{')][-1])
print(round(d['ms_per_step'],2), d['config']['step_parts_ms_rank0'], d['config']['recount_ms'], d['config']['sites_counted'], d['config']['step1_candidates'])
"
done
