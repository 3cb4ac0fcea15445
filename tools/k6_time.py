#!/usr/bin/env python3
"""Development aid: time pf_scan_vote (K6, S3P:328-376) at the reference's size (5 x 5^3 pose
hypotheses x 500 points, 200x200x80 grid) and at a widened lattice, with the oracle beside it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
from oracle import binding as O

dims = (200, 200, 80)
grid = (synth.make_room(dims, seed=4) > 50).astype(np.uint8)
rng = np.random.default_rng(0)
pos = np.array([5.0, 5.0, 1.5]); zyx = np.array([0.3, 0.01, -0.02])
occ = np.argwhere(grid > 0)
h = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                                    grid_z=dims[2], max_points=1 << 16))
h.write_map(0, grid.reshape(-1))
for n_pts, n_yaw, n_t in ((500, 5, 5), (19200, 21, 9)):
    wp = occ[rng.integers(0, len(occ), n_pts)] * 0.05 + rng.normal(0, 0.02, (n_pts, 3))
    pts = (wp - pos) @ np.array(O.rot_zyx(*zyx)).reshape(3, 3)
    kw = dict(n_yaw=n_yaw, yaw_k0=-(n_yaw // 2) * 3, yaw_kstep=3, n_t=n_t, length=0.05 * n_t)
    sums, best = h.scan_vote(pts, pos, zyx, **kw)
    h.profile(True, [capi.PF_K_VOTE])
    t0 = time.perf_counter()
    for _ in range(20):
        h.scan_vote(pts, pos, zyx, **kw)
    call = (time.perf_counter() - t0) / 20
    ms, n = h.kernel_time(capi.PF_K_VOTE)
    h.profile(False)
    t0 = time.perf_counter(); so, bo = O.scan_vote(grid, 0.05, pts, pos, zyx, **kw); cpu = time.perf_counter() - t0
    assert np.array_equal(sums, so) and best == bo
    look = n_yaw * n_t ** 3 * n_pts
    print(f"{n_yaw * n_t ** 3} hypotheses x {n_pts} points = {look:.2e} look-ups: kernels {ms / n * 1e3:.1f} us "
          f"({look / (ms / n * 1e-3):.2e} look-ups/s), whole call with upload and read-back {call * 1e6:.0f} us, "
          f"oracle on one core {cpu * 1e3:.1f} ms; equal")
h.close()
