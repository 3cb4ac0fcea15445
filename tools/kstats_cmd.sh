#!/bin/bash
# Development aid: per-kernel stats of any python script. usage: bash tools/kstats_cmd.sh <tag> <script.py> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-30s %5s %9.1f" % (r["Name"].split("(")[0][:30], r["Calls"], float(r["AverageNs"])))
print(f)
PY
