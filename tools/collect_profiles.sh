#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 summaries behind bench.py's roofline numbers.  Each counter gets its own pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains).  Outputs under gpurun_out/prof/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 5 --warmup 2 > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.out 2> $O/write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -o calib -- ./tools/pmc_calib.bin > $O/calib.out 2> $O/calib.err
find $O -name "*.csv" | head -20
