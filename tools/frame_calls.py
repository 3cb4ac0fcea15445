#!/usr/bin/env python3
"""Development aid: host time per library call of the per-frame loop (slam_amd.frontend.FrameLoop on the 48 recorded
frames of tests/golden/frames.npz), with the cloud kept in HBM and with the host hand-off; what is left of a frame's wall
time is the Python of the loop itself.   python tools/frame_calls.py"""
import os, sys, time, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from slam_amd import capi
from slam_amd import frontend
from slam_amd.frontend import FrameLoop
acc = collections.defaultdict(float)
def wrap(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(obj, name, w)
for n in ("draw_indices", "vote_hypothesis", "vote_epilogue", "update_realtime_pos"):
    wrap(capi, n)
for n in ("depth_to_cloud_device", "scan_vote_device", "ring_update_device", "set_scan_device", "ring_paint", "update_map",
          "depth_to_cloud", "scan_vote", "ring_update", "set_scan"):
    wrap(capi.Handle, n)
g = np.load("tests/golden/frames.npz"); F = int(g["frames"])
images = [np.repeat(np.repeat(g["pooled"][f], 4, 0), 4, 1).astype("<u2").tobytes() for f in range(F)]
for dev in (True, False, True):
    loc = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=200, grid_y=200, grid_z=80, local_x=0, local_y=0, local_z=0, max_beams=64))
    mapper = capi.Handle(capi.default_config(n_particles=1, max_beams=1 << 18))
    loc.fill_maps(0); loc.ring_init()
    fl = FrameLoop(loc, mapper, replay=True, device_cloud=dev)
    acc.clear()
    t0 = time.perf_counter()
    for f in range(F):
        fl.frame(images[f], g["gyr"][f], int(g["seed_vote"][f]), int(g["seed_rays"][f]))
    mapper.synchronize(); loc.synchronize()
    wall = time.perf_counter() - t0
    print("device" if dev else "host", f"{wall / F * 1e6:.1f} us/frame", {k: round(v / F * 1e6, 1) for k, v in acc.items()}, "sum", round(sum(acc.values()) / F * 1e6, 1))
    fl.close(); loc.close(); mapper.close()
