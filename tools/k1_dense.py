#!/usr/bin/env python3
"""K1 (measure on the shared grid) on a scan of the reference's own shape: the occupied cells (v > 20) of the
local map its class Slam holds after the first 8 frames of tests/golden/frames.npz (3615 beams in 765 local
columns: walls as z runs, the floor as single cells), P particles around the sensor in a 300 x 300 x 80 room
grid (S3P:40-42). Scores are checked against the oracle before anything is timed.
    python tools/k1_dense.py [P]        (SLAM_PF_RUNS=0: without the run groups)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from slam_amd import capi, synth
from oracle import binding as O
import golden_util as G
from oracle_backend import OracleLocHandle, OracleMapper

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(os.environ.get("K1_REPS", 20))
mapper = OracleMapper()
G.check_frames(os.path.join(ROOT, "tests", "golden"), OracleLocHandle(), mapper, None, frames=8)
local = mapper.read_local_map()
beams = capi.snapshot_beams(local.reshape(200, 200, 80), (100, 100, 30))
occ = beams[3] > 20
dims = (300, 300, 80)
grid = synth.make_room(dims, 3)
rng = np.random.default_rng(2)
def cloud(k):
    r = np.random.default_rng(100 + k)
    return (7.5 + r.normal(0, 0.03, P), 7.5 + r.normal(0, 0.03, P), 1.5 + r.normal(0, 0.015, P), 0.3 + r.normal(0, 0.02, P))
cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                          local_x=0, local_y=0, local_z=0, max_beams=len(beams[0]))
h = capi.Handle(cfg)
x, y, z, yaw = cloud(0)
h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_beams(*beams)
h.measure(0); h.synchronize()
s, c = h.get_scores()
pick = np.arange(0, P, max(1, P // 64))[:64]
g1 = grid.reshape(-1).copy()
so, co = O.measure(O.ref_grid(*dims), x[pick], y[pick], z[pick], yaw[pick], g1, 0, *beams, 0)
assert np.array_equal(s[pick], so) and np.array_equal(c[pick], co), "scores differ from the oracle"
h.profile(True, [capi.PF_K_MEASURE, capi.PF_K_ORDER])
for k in range(reps):
    h.predict(*cloud(k + 1)); h.measure(0)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
mo, no = h.kernel_time(capi.PF_K_ORDER)
Mo = int(occ.sum())
t = ms / n * 1e-3
print(f"K1 on the reference-shaped scan: {P} particles x {Mo} occupied beams ({len(beams[0])} non-zero cells in the list), "
      f"300x300x80 grid: {t * 1e6:.2f} us per launch (order+reduce {mo / no * 1e3:.2f} us), {P * Mo / t:.3e} evals/s = "
      f"{P * Mo / t / 8e12:.3f} of the 1 B/eval roofline   env: "
      + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SLAM_PF")), flush=True)
h.close()
