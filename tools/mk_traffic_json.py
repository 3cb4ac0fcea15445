#!/usr/bin/env python3
"""Turns the counter passes of tools/pmc_k1.sh (gpurun_out/pmc_<tag>/) into profiles/<out>_k1_traffic.json and a text
summary profiles/<out>_pmc_k1_tile.txt.   usage: python tools/mk_traffic_json.py <tag> <out-prefix>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_measure" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
per = {k: v / n for k, (v, n) in acc.items()}
with open(os.path.join(ROOT, "profiles", f"{out}_pmc_k1_tile.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes of tools/pmc_k1.sh (each pass its own run of tools/k1_only.py: 10 000 particles x 4096 beams, 256^3 grid), k_measure_tile, per launch\n")
    for k in sorted(per): f.write(f"{k:40s} {per[k]:18.1f}\n")
fetch, write = per["FETCH_SIZE"], per["WRITE_SIZE"]
j = {"kernel": "k_measure_tile", "workload": "C2: 10000 particles x 4096 beams, 256^3 grid",
     "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
     "traffic_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
     "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (tools/pmc_k1.sh), per launch; FETCH_SIZE doubled as "
             "MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads (the window loads are 256 contiguous bytes per wave); "
             "the figure includes the kernel's register spills to scratch memory",
     "source": f"profiles/{out}_pmc_k1_tile.txt",
     "TCC_HIT_per_launch": per.get("TCC_HIT_sum"), "TCC_MISS_per_launch": per.get("TCC_MISS_sum"),
     "l2_hit_rate": round(per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), 4)}
json.dump(j, open(os.path.join(ROOT, "profiles", f"{out}_k1_traffic.json"), "w"), indent=1)
print(json.dumps(j))
