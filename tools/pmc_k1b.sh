#!/bin/bash
# Second counter set for K1: TA/TD/TCP stall breakdown.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$1
mkdir -p $out
run() { n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -- python3 tools/k1_only.py > $out/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $out/$n.log; return 1; }
}
run s1 TD_TD_BUSY_sum TD_TC_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run s2 TA_TA_BUSY_sum TA_BUSY_avr TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE &&
run s3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "k_measure" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:44s} per-launch {v / n:18.1f}  (launches {n})")
PY
