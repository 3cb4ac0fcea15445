#!/bin/bash
# Development aid: per-kernel averages of one rank's rehearsed step at a world size.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kw$w -- python3 tools/rehearse_world.py --world $w > gpurun_out/kw$w.log 2>&1 || { tail -3 gpurun_out/kw$w.log; exit 1; }
f=$(find gpurun_out/kw$w -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:14]:
    if int(r["Calls"]) > 100:
        print(f'{r["Name"][:34]:36s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1000:8.2f} us')
        tot += float(r["AverageNs"]) / 1000
print("sum", round(tot, 1))
PY
