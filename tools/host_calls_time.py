#!/usr/bin/env python3
"""Development aid: host time of the per-step upload calls (no stream waits inside the loops)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import capi, synth
P = int(os.environ.get("K1_P", 10000)); M = int(os.environ.get("K1_M", 4096)); G = 256
sc = synth.scene((G, G, G), P, M, seed=0)
h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                                    local_x=0, local_y=0, local_z=0, max_beams=M, max_points=len(sc["points"])))
h.write_map(0, sc["grid"])
x, y, z, yaw = sc["x"], sc["y"], sc["z"], sc["yaw"]
u = capi.draw_uniform(3, P)
def t(name, fn, n=200):
    fn(); h.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0
    h.synchronize()
    print(f"{name:12s} {el / n * 1e6:8.1f} us per call (host)", flush=True)
# a process that has torch loaded runs its first second or so at a fraction of its steady rate (the
# settle phase of bench.py is there for the same reason): let that pass before timing anything
t_end = time.perf_counter() + 1.5
while time.perf_counter() < t_end:
    h.predict(x, y, z, yaw); h.set_beams(*sc["beams"])
h.synchronize()
t("predict", lambda: h.predict(x, y, z, yaw))
t("set_beams", lambda: h.set_beams(*sc["beams"]))
t("set_scan", lambda: h.set_scan(sc["points"], sc["ray_idx"]))
t("set_weights", lambda: h.set_weights(u))
t("set_best_index (binding overhead of a trivial call)", lambda: h.set_best_index(0), 2000)
h.close()
