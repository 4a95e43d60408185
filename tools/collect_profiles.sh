#!/bin/bash
# Run on the GPU box (gpurun): the rocprofv3 summaries behind bench.py's roofline numbers.  Every counter set gets its
# own pass (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with trace domains).  Raw output
# goes to gpurun_out/prof/; tools/summarize_profiles.py condenses it into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4"
echo trace;  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-reads 0 --no-recount --no-c4 > $O/bench_under_rocprof.json 2> $O/trace.err
echo fetch;  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- $B > $O/fetch.out 2> $O/fetch.err
echo write;  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- $B > $O/write.out 2> $O/write.err
echo rdreq;  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq -o bench -- $B > $O/rdreq.out 2> $O/rdreq.err
echo calib;  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -o calib -- ./tools/pmc_calib.bin > $O/calib.out 2> $O/calib.err
echo done; find $O -name "*.csv" | wc -l
