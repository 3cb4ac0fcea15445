#!/bin/bash
# Development aid: rocprofv3 kernel stats of the per-frame leg alone (bench.py --frame-only); the summary lands in
# gpurun_out/fprof_kernel_stats.csv (committed as profiles/r3_frame_kernel_stats.csv).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fprof
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fprof -- python3 bench.py --frame-only > gpurun_out/fprof.log 2>&1 || { tail -5 gpurun_out/fprof.log; exit 1; }
f=$(find gpurun_out/fprof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/fprof_kernel_stats.csv
python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:50]:52s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1000:9.2f} us {r["Percentage"]}')
P
