#!/bin/bash
# Round-end evidence: the default bench line, then the same command under rocprofv3 --kernel-trace --stats.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo bench failed; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py > gpurun_out/${tag}_bench_profiled.json 2> gpurun_out/${tag}_prof.err || { echo profiled bench failed; tail -5 gpurun_out/${tag}_prof.err; exit 1; }
f=$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -20 gpurun_out/${tag}_kernel_stats.csv
