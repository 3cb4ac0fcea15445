#!/bin/bash
# Counter passes for K2 on BASELINE configs[3]'s sparse scan (2048 beams, private 128^3 maps; 5 000 particles by default: the bytes per particle are the same and the profiler crashed on the 105 GB of the full size): HBM-side
# bytes per launch of k_measure_fuse_wave, each pass its own run.   usage: bash tools/pmc_k2_sparse.sh <tag> [particles]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$1; NP=${2:-5000}
mkdir -p $out
run() { n=$1; shift
  timeout -s KILL 280 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 tools/c4_check.py $NP > $out/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $out/$n.log; return 1; }
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
grep "K2 roofline" $out/fetch.log
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_measure_fuse" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:32s} per-launch {v / n:16.1f}  (launches {n})")
PY
