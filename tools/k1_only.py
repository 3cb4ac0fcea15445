#!/usr/bin/env python3
"""Development aid for counter passes: runs only pf_measure (K1) at the C2 size a few times.
    rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/k1_only.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth  # noqa: E402

P = int(os.environ.get("K1_P", 10000)); M = int(os.environ.get("K1_M", 4096))
G = int(os.environ.get("K1_G", 256)); reps = int(os.environ.get("K1_REPS", 10))
sc = synth.scene((G, G, G), P, M, seed=0)
cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                          local_x=0, local_y=0, local_z=0, max_beams=max(M, 1))
h = capi.Handle(cfg)
h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); h.set_beams(*sc["beams"])
import numpy as np  # noqa: E402
for k in range(reps):
    # new poses every time, same cloud statistics: the lane ordering runs and sizes its cells by the last cloud
    rng = np.random.default_rng(k)
    tp = sc["true_pose"]
    h.predict(tp[0] + rng.normal(0, 0.03, P), tp[1] + rng.normal(0, 0.03, P), tp[2] + rng.normal(0, 0.015, P),
              tp[3] + rng.normal(0, 0.02, P))
    h.measure(0)
h.synchronize()
s, c = h.get_scores()
print("k1_only done", int(s[0]), int(c[0]))
h.close()
