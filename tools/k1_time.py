#!/usr/bin/env python3
"""Development aid: K1's launch time (HIP events) at a given size, for whatever SLAM_PF_* knobs are set.
    K1_P=10000 K1_M=4096 K1_G=256 python tools/k1_time.py"""
import os
import sys

import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth  # noqa: E402

P = int(os.environ.get("K1_P", 10000)); M = int(os.environ.get("K1_M", 4096))
G = int(os.environ.get("K1_G", 256)); reps = int(os.environ.get("K1_REPS", 20))
sc = synth.scene((G, G, G), P, M, seed=0)
cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                          local_x=0, local_y=0, local_z=0, max_beams=max(M, 1))
h = capi.Handle(cfg)
h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); h.set_beams(*sc["beams"])
h.measure(0); h.synchronize()
s0, c0 = h.get_scores()
h.profile(True, [capi.PF_K_MEASURE, capi.PF_K_ORDER])
for k in range(reps):
    # new poses every time (the ordering runs), same cloud statistics
    rng = np.random.default_rng(k)
    tp = sc["true_pose"]
    h.predict(tp[0] + rng.normal(0, 0.03, P), tp[1] + rng.normal(0, 0.03, P), tp[2] + rng.normal(0, 0.015, P),
              tp[3] + rng.normal(0, 0.02, P))
    h.measure(0)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
mo, no = h.kernel_time(capi.PF_K_ORDER)
Mr = len(sc["beams"][0])
print(f"K1 {ms / n * 1e3:7.2f} us  order+reduce {mo / n * 1e3:6.2f} us  ({P} x {Mr}, {G}^3) "
      f"-> {P * Mr / (ms / n * 1e-3) / 8e12:.3f} of roofline   env: "
      + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SLAM_PF")), flush=True)
h.close()
