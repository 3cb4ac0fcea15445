"""MI355X-native particle-filter hot path of Gabrieleenx/SLAM.

The product is slam_amd/csrc/libslam_pf.so behind include/slam_pf.h; this package is the thin
Python plumbing around it (ctypes binding, host-driver mirror, multi-GPU sharding).
"""
from . import capi  # noqa: F401
