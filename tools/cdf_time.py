#!/usr/bin/env python3
"""Development aid: run pf_sequential_cdf at a few sizes (read the kernel times from rocprofv3's trace)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import capi
cfg = capi.default_config(device=0, n_particles=64, n_particles_global=64, particle_offset=0,
                          map_mode=capi.PF_MAP_SHARED, grid_x=16, grid_y=16, grid_z=16,
                          local_x=0, local_y=0, local_z=0, max_beams=8, max_points=8)
h = capi.Handle(cfg)
rng = np.random.default_rng(0)
for n in (8000, 80000, 1000000):
    w = rng.random(n) + 1e-3; w /= w.sum()
    for how in (1, 1, 2, 2, 0):
        if how == 0 and n > 100000:
            continue
        h.sequential_cdf(w, how)
h.close()
