#!/usr/bin/env python3
"""Development aid: K1 time at the C2 size for several particle-order keys
(SLAM_PF_ORDER=qa,qx,qy,mode,bits; see order_key in pf_kernels.hip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
G = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sc = synth.scene((G, G, G), P, M, seed=0)
M = len(sc["beams"][0])
cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                          local_x=0, local_y=0, local_z=0, max_beams=max(M, 1))
h = capi.Handle(cfg)
h.write_map(0, sc["grid"]); h.set_beams(*sc["beams"])
ref = None
keys = sys.argv[4:] or ["500,2,4,0,5", "300,3,3,1,5", "300,3,3,1,4", "200,2,2,1,4", "400,4,4,1,4", "150,3,3,1,4",
                        "300,2,2,1,4", "200,3,3,1,4"]
for key in keys:
    os.environ["SLAM_PF_ORDER"] = key
    h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
    for _ in range(3):
        h.measure(0)
    h.profile(True)
    for _ in range(30):
        h.measure(0)
    ms, n = h.kernel_time(capi.PF_K_MEASURE)
    h.profile(False)
    s, c = h.get_scores()
    if ref is None:
        ref = (s.copy(), c.copy())
    same = np.array_equal(s, ref[0]) and np.array_equal(c, ref[1])
    us = ms / n * 1e3
    print(f"order {key:>18s}  K1 {us:7.2f} us  frac8TB {(P * M + 56 * P + 13 * M) / us * 1e6 / 8e12:.3f}  same={same}", flush=True)
h.close()
