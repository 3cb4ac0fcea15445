#!/bin/bash
# Counter passes for K1 (each pass its own run, within the per-block slot limits).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$1
mkdir -p $out
run() { # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 tools/k1_only.py > $out/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $out/$n.log; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY &&
run mem TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_measure" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:40s} per-launch {v / n:18.1f}  (launches {n})")
PY
