#!/bin/bash
# TEST INFRASTRUCTURE: builds oracle/_ref/libs3p_ref.so from the reference's own text, where it
# lies under /root/reference (never copied into this repository; the output directory is
# git-ignored). Runs only where /root/reference exists (the build container).
#
# What is compiled: the reference's ROS-free part, streamed to g++ by line range:
#   S3P:12-110   system includes, globals, struct map_particles, index helpers
#   S3P:731-962  rot_matrix, rotate, reset_local_map, resample(), particle_map()
# followed by oracle/ref_harness_tail.cpp (accessors; same translation unit).
# What is NOT compiled: S3P:1-11 (ROS headers, absent from this image), class Slam
# (S3P:112-729: its members are ROS publishers/messages) and the node shell (S3P:964-1118).
# No stand-in for any ROS header, library or type is written: the hot path of
# particle_map()/resample() does not use any. build_local_map / scan_localization are members
# of class Slam and therefore unbuildable here; their oracle is the restatement only.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref=/root/reference/Slam_3d_particle.cpp
[ -f "$ref" ] || { echo "build_ref.sh: $ref not present; skipping (GPU box uses fixtures)"; exit 0; }
mkdir -p "$here/_ref"
( sed -n '12,110p;731,962p' "$ref"; cat "$here/ref_harness_tail.cpp" ) | \
  g++ -x c++ -std=gnu++14 -O2 -fPIC -shared -w -include "$here/ref_pin.h" \
      -o "$here/_ref/libs3p_ref.so" -
echo "built $here/_ref/libs3p_ref.so"
