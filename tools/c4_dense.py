#!/usr/bin/env python3
"""K2 (measure + fuse on per-particle maps) on a DENSE local map of the reference's own shape: the
local map the reference's class Slam had has after the first 8 frames of tests/golden/frames.npz, i.e. what its first
particle_map() fuses (200 x 200 x 80 int8, ~139 k non-zero cells: walls as full columns plus the free cells of
2400 rays per frame), fused into P per-particle 300 x 300 x 80 maps (S3P:40-42, 7.2 MB each).
    python tools/c4_dense.py [P]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from slam_amd import capi
from oracle import binding as O

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
import golden_util as G
from oracle_backend import OracleLocHandle, OracleMapper
mapper = OracleMapper()
G.check_frames(os.path.join(ROOT, "tests", "golden"), OracleLocHandle(), mapper, None, frames=8)   # 8 callback() frames
local = mapper.read_local_map()
beams = capi.snapshot_beams(local.reshape(200, 200, 80), (100, 100, 30))
M = len(beams[0]); Mo = int((beams[3] > 20).sum())
rng = np.random.default_rng(2)
x = 7.5 + rng.normal(0, 0.06, P); y = 7.5 + rng.normal(0, 0.06, P); z = 1.5 + rng.normal(0, 0.03, P)
yaw = 0.3 + rng.normal(0, 0.05, P)
world = np.full(300 * 300 * 80, 50, np.uint8)
h = capi.Handle(capi.default_config(n_particles=P, max_beams=M))
h.broadcast_map(world); h.predict(x, y, z, yaw); h.set_beams(*beams)
h.measure(1); h.synchronize()
s, c = h.get_scores()
pick = np.arange(0, P, max(1, P // 8))[:8]
maps = np.tile(world.reshape(1, -1), (len(pick), 1))
so, co = O.measure(O.ref_grid(), x[pick], y[pick], z[pick], yaw[pick], maps.reshape(-1), maps.shape[1], *beams, 1)
assert np.array_equal(s[pick], so) and np.array_equal(c[pick], co)
for k, i in enumerate(pick[:3]):
    assert np.array_equal(h.read_map(int(i)), maps[k])
h.profile(True)
for _ in range(3):
    h.measure(1)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
t = ms / n * 1e-3
alg = 2.0 * P * M
print(f"K2 measure+fuse: {P} particles x {M} beams ({Mo} occupied), 300x300x80 maps: {t * 1e3:.3f} ms per launch, "
      f"{P * M / t:.3e} evals/s, algorithmic {alg / t / 1e12:.2f} TB/s = {alg / t / 8e12 * 100:.1f} % of 8 TB/s", flush=True)
h.profile(False)
h.measure(0); h.synchronize(); h.profile(True)
for _ in range(3):
    h.measure(0)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
print(f"K1 measure only (occupied beams, per-particle maps): {ms / n:.3f} ms per launch")
h.close()
