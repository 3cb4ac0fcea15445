#!/bin/bash
# Development aid: per-kernel stats of the per-frame leg. usage: bash tools/kstats_frame.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --frame-only "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-40s %6s %9.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])))
print(f)
PY
