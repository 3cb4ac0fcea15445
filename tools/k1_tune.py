#!/usr/bin/env python3
"""Development aid: time pf_measure (K1) alone for several beam-chunk counts and check that
the fast (fp32-filter) kernel and the fp64-only kernel give identical scores."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth  # noqa: E402


def run(P, M, G, chunks, exact, reps=50, sc=None, variant=0):
    os.environ["SLAM_PF_CHUNKS"] = str(chunks)
    os.environ["SLAM_PF_MEASURE_VARIANT"] = str(variant)
    os.environ["SLAM_PF_EXACT_MEASURE"] = "1" if exact else "0"
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G,
                              grid_z=G, local_x=0, local_y=0, local_z=0, max_beams=max(M, 1))
    h = capi.Handle(cfg)
    h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
    h.set_beams(*sc["beams"])
    for _ in range(5):
        h.measure(0)
    h.profile(True)
    for _ in range(reps):
        h.measure(0)
    ms, n = h.kernel_time(capi.PF_K_MEASURE)
    s, c = h.get_scores()
    h.close()
    return ms / n * 1e3, s, c


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    sc = synth.scene((G, G, G), P, M, seed=0)
    M = len(sc["beams"][0])
    ref = None
    for exact, variant in ((1, 0), (0, 1), (0, 0)):
        for chunks in ((0,) if exact else (0, 25, 51)):
            us, s, c = run(P, M, G, chunks, exact, sc=sc, variant=variant)
            if ref is None:
                ref = (s, c)
            same = np.array_equal(s, ref[0]) and np.array_equal(c, ref[1])
            print(f"exact={exact} variant={variant} chunks={chunks:4d} K1={us:8.2f} us  evals/s={P * M / us * 1e6:.3e} "
                  f"frac8TB={(P * M + 56 * P + 13 * M) / us * 1e6 / 8e12:.3f} same={same}", flush=True)
