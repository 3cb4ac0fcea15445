#!/usr/bin/env python3
"""Development probe: K1 time when every particle has the SAME pose (one cache access per
wave-gather) against the normal cloud. Separates VALU/latency cost from TA/TCP cost."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
for label, f in (("cloud", 1.0), ("tight(0.1x)", 0.1), ("identical", 0.0)):
    for exact in (0, 1):
        os.environ["SLAM_PF_EXACT_MEASURE"] = str(exact)
        cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G,
                                  grid_z=G, local_x=0, local_y=0, local_z=0, max_beams=M)
        h = capi.Handle(cfg)
        tp = sc["true_pose"]
        x = tp[0] + (sc["x"] - tp[0]) * f; y = tp[1] + (sc["y"] - tp[1]) * f
        z = tp[2] + (sc["z"] - tp[2]) * f; yaw = tp[3] + (sc["yaw"] - tp[3]) * f
        h.write_map(0, sc["grid"]); h.predict(x, y, z, yaw); h.set_beams(*sc["beams"])
        for _ in range(5):
            h.measure(0)
        h.profile(True)
        for _ in range(30):
            h.measure(0)
        ms, n = h.kernel_time(capi.PF_K_MEASURE)
        print(f"{label:12s} exact={exact} K1={ms / n * 1e3:7.2f} us", flush=True)
        h.close()
