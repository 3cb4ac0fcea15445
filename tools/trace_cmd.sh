#!/bin/bash
# Development aid: every launch of kernels matching <pattern> with its duration. usage: bash tools/trace_cmd.sh <tag> <pattern> <script.py> [args...]
tag=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o $tag -- python3 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        print("%-30s %10.1f us" % (r["Kernel_Name"].split("(")[0][:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
