#!/bin/bash
# Development aid: per-kernel averages of one rank's rehearsed step at the scale of BASELINE configs[2]
# (125 k particles per GPU x 16 k beams, W ranks).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1
rm -rf gpurun_out/kc3_$w
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc3_$w -- python3 tools/rehearse_world.py --world $w --particles 125000 --beams 16384 --steps 40 --warmup 5 --fresh 4 > gpurun_out/kc3_$w.log 2>&1 || { tail -3 gpurun_out/kc3_$w.log; exit 1; }
f=$(find gpurun_out/kc3_$w -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    if int(r["Calls"]) >= 40:
        print(f'{r["Name"][:40]:42s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1000:9.2f} us')
PY
