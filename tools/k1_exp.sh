#!/bin/bash
# Development aid: K1's time for the tile kernel, its timing-only variants (tools/_exp/*.so), the brick kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout -k 10 120 env "$@" python3 tools/k1_time.py 2>&1 | grep "^K1" || echo "failed: $*"; }
run A=tile
run SLAM_PF_K1=col
for e in "$@"; do run SLAM_PF_LIB=$PWD/tools/_exp/$e.so; done
