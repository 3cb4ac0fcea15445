#!/bin/bash
# TEST INFRASTRUCTURE: builds oracle/_ref/libs3p_ref.so from the reference's own text, where it
# lies under /root/reference (never copied into this repository; the output directory is
# git-ignored). Runs only where /root/reference exists (the build container).
#
# What is compiled: the reference's ROS-free text, streamed to g++ by line range:
#   S3P:12-110   system includes, globals, struct map_particles, index helpers
#   (harness)    `class Slam{ public:`          -- replaces S3P:112-114, nothing else
#   S3P:115-146, 150-151   class Slam's plain data members (depth arrays, tables, cloud, diffs)
#   (harness)    `Slam(){` ... `}`              -- around S3P:157-162, the constructor's two
#                                                  table loops (the reference's ctor takes a
#                                                  ros::NodeHandle* and advertises a topic)
#   S3P:258-729  everything behind `private:` (S3P:257, not streamed, so the members stay
#                reachable): scan_localization, scan_localization_map_update, rotate_point,
#                Translate_point, resize_depth_to_array, calc_mean_16, img_to_cloud, norm,
#                build_local_map, update_realtime_pos and the closing brace of the class
#   S3P:731-962  rot_matrix, rotate, reset_local_map, resample(), particle_map()
# followed by oracle/ref_harness_tail.cpp and oracle/ref_harness_slam.cpp (accessors and thin
# callers; same translation unit).
# What is NOT compiled: S3P:1-11 (ROS headers, absent from this image), S3P:147 and 153
# (ROS-typed members), S3P:155-156, 163-166 (ROS constructor), S3P:168-255 (callback,
# pub_pose_func, filter_callback, sync_filter: ROS message plumbing) and the node shell
# (S3P:964-1118). No stand-in for any ROS header, library or type is written: none of the
# streamed lines names one.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref=/root/reference/Slam_3d_particle.cpp
[ -f "$ref" ] || { echo "build_ref.sh: $ref not present; skipping (GPU box uses fixtures)"; exit 0; }
mkdir -p "$here/_ref"
( sed -n '12,110p' "$ref"
  echo 'class Slam{ public:'
  sed -n '115,146p;150,151p' "$ref"
  echo 'Slam(){'
  sed -n '157,162p' "$ref"
  echo '}'
  sed -n '258,729p;731,962p' "$ref"
  cat "$here/ref_harness_tail.cpp" "$here/ref_harness_slam.cpp" ) | \
  g++ -x c++ -std=gnu++14 -O2 -fPIC -shared -w -include "$here/ref_pin.h" \
      -o "$here/_ref/libs3p_ref.so" -
echo "built $here/_ref/libs3p_ref.so"
