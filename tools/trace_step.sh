#!/bin/bash
# Run on the GPU box (gpurun): kernel trace of a short bench run -> the timeline of its last step (tools/timeline.py) in gpurun_out/timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_q; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4 > $O/bench.json 2> $O/trace.err || { echo trace failed; tail -5 $O/trace.err; exit 1; }
python3 tools/timeline.py $O > gpurun_out/timeline.txt
find $O -name "*kernel_trace.csv" -delete
tail -80 gpurun_out/timeline.txt
