#!/bin/bash
# Development aid: K1 launch time against the number of beam chunks. usage: bash tools/chunks_sweep.sh 34 40 51 64
for c in "$@"; do
  SLAM_PF_CHUNKS=$c timeout -k 10 120 python bench.py --no-cpu-baseline --no-fuse-kernels --settle-seconds 0.3 > gpurun_out/chunks_$c.json 2>/dev/null || exit 1
  echo -n "chunks $c: "; python tools/print_bench.py gpurun_out/chunks_$c.json
done
