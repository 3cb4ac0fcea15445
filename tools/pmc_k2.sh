#!/bin/bash
# Development aid: counter passes for the K2 kernel on the dense scene (each pass its own run).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$1
mkdir -p $out
run() { n=$1; shift
  timeout -s KILL 250 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 tools/c4_dense.py ${K2_P:-4000} > $out/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $out/$n.log; return 1; }
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS
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
