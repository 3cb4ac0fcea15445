#!/usr/bin/env python3
"""One GPU's share of BASELINE configs[2] and configs[4] (the driver runs the 8-GPU jobs; here the
per-GPU problem is checked and timed on one MI355X):
  c3: 125 000 particles x 16 384-beam scan, 256^3 shared grid   (1 M particles / 8 GPUs)
  c5: 12 500 particles x 131 072-beam scan, 1024^3 shared grid  (100 k particles / 8 GPUs)
Beams are the local voxels of randomly chosen wall cells seen from the true pose (no ray trace:
a 1024^3 trace in numpy takes minutes). A sample of particles is checked against the oracle.
    python tools/c35_check.py c3 | c5
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
from oracle import binding as O

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
P, M, G = (125000, 16384, 256) if which == "c3" else (12500, 131072, 1024)
if len(sys.argv) > 2:
    P = int(sys.argv[2])
dims = (G, G, G)
rng = np.random.default_rng(11)
t0 = time.time()
grid = synth.make_room(dims, seed=1)
res = 0.05
true_vox = np.array([G / 2 + 0.3, G / 2 - 0.2, G / 2 + 0.1]); yaw0 = 0.35
# wall cells within range, as local voxels of the true pose (rotate by -yaw, round), distinct
occ = np.argwhere(grid[::2, ::2, ::2] == 100) * 2 if G > 256 else np.argwhere(grid == 100)
pick = occ[rng.choice(len(occ), min(len(occ), 3 * M), replace=False)].astype(np.float64)
d = pick - true_vox
c, s = np.cos(-yaw0), np.sin(-yaw0)
loc = np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1], d[:, 2]], 1)
loc = np.unique(np.rint(loc).astype(np.int32), axis=0)[:M]
bx, by, bz = (np.ascontiguousarray(loc[:, k]) for k in range(3))
bv = np.full(len(bx), 50, np.int8)
M = len(bx)
x = true_vox[0] * res + rng.normal(0, 0.03, P); y = true_vox[1] * res + rng.normal(0, 0.03, P)
z = true_vox[2] * res + rng.normal(0, 0.015, P); yaw = yaw0 + rng.normal(0, 0.02, P)
print(f"{which}: {P} particles x {M} beams, {G}^3 grid; scene built in {time.time() - t0:.1f} s", flush=True)

h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                                    local_x=0, local_y=0, local_z=0, max_beams=M))
h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_weights(np.full(P, 1.0 / P)); h.set_beams(bx, by, bz, bv)
h.measure(0); h.synchronize()
h.profile(True)
for _ in range(5):
    h.measure(0)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
k1 = ms / n * 1e-3
nbytes = P * M + 56 * P + 13 * M
print(f"K1: {k1 * 1e3:.3f} ms per launch, {P * M / k1:.3e} evals/s, {nbytes / k1 / 1e12:.2f} TB/s algorithmic "
      f"= {nbytes / k1 / 8e12 * 100:.1f} % of 8 TB/s", flush=True)
sg, cg = h.get_scores()
pickp = np.arange(0, P, max(1, P // 24))
so, co = O.measure(O.ref_grid(*dims), x[pickp], y[pickp], z[pickp], yaw[pickp], grid.reshape(-1), 0, bx, by, bz, bv, 0)
assert np.array_equal(sg[pickp], so) and np.array_equal(cg[pickp], co), "scores differ from the oracle"
u = capi.draw_uniform(4, P)
# one untimed weights + resample first (the first launch of a kernel loads its code), then the same
# state again for the step that is timed and checked
h.profile(False)
h.weights(); h.resample(u)
h.predict(x, y, z, yaw); h.set_weights(np.full(P, 1.0 / P)); h.measure(0)
h.profile(True)
h.weights()
t0 = time.time(); slots, best, nohit, amb = h.resample(u); t1 = time.time() - t0
w = np.full(P, 1.0 / P); O.weights(w, sg, int(cg[-1])); _, sa, ob, onh = O.resample(w, u, 0)
assert np.array_equal(slots, sa) and best == ob and nohit == onh, "resampling differs from the oracle"
for name, kid in (("weights", capi.PF_K_WEIGHTS), ("resample", capi.PF_K_RESAMPLE)):
    ms, n = h.kernel_time(kid)
    print(f"{name}: {ms / max(n, 1) * 1e3:.1f} us", flush=True)
print(f"{which} check ok (scores of {len(pickp)} particles and all {P} resampled slots equal the oracle; ambiguous draws {amb})")
h.close()
