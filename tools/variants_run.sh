#!/bin/bash
# Run on the GPU box: bench.py's step under every library variant in longsom_amd/lib/variants/*.so (tools: built by hand with other -D flags)
# and under the LSG_GRID_TD values given on the command line; the shipped library last.
cd "$GRAFT_REPO_ROOT"
cp longsom_amd/lib/liblongsom_hip.so /tmp/shipped.so
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4"
one() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 $B > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err || echo "$name failed"
  python3 - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/v_%s.json" % n).read().strip().splitlines()[-1])
    print(n, "step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]), d["config"]["step_parts_ms_rank0"], flush=True)
except Exception as e:
    print(n, "ERR", e)
PY
}
for v in longsom_amd/lib/variants/*.so; do
  n=$(basename $v .so)
  cp $v longsom_amd/lib/liblongsom_hip.so
  for g in "$@"; do one ${n}_g$g LSG_GRID_TD=$g LSG_GRID_TW=$g; done
done
cp /tmp/shipped.so longsom_amd/lib/liblongsom_hip.so
for g in "$@"; do one shipped_g$g LSG_GRID_TD=$g LSG_GRID_TW=$g; done
