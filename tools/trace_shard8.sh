cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace8; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -o sp -- python3 tools/shard_perf.py 1e7 8 > $O/sp.log 2> $O/sp.err || { echo failed; tail -5 $O/sp.err; exit 1; }
python3 tools/timeline.py $O > gpurun_out/timeline8.txt
find $O -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/timeline8.txt
