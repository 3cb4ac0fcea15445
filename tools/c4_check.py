#!/usr/bin/env python3
"""configs[3] sanity run: FastSLAM-style per-particle 128^3 maps, 50k particles (104.9 GB of
maps on one MI355X). One measure+fuse, weights, resample (map slabs move with the particles);
a sample of particles is checked against the oracle. Prints timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
from oracle import binding as O

P = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dims = (128, 128, 128)
sc = synth.scene(dims, P, M, seed=5)
bx, by, bz, bv = sc["beams"]
rng = np.random.default_rng(1)
bv = rng.choice(np.array([-30, -3, 10, 30, 50], np.int8), len(bx))        # hits and misses
t0 = time.time()
h = capi.Handle(capi.default_config(n_particles=P, grid_x=128, grid_y=128, grid_z=128,
                                    local_x=0, local_y=0, local_z=0, max_beams=len(bx)))
h.broadcast_map(sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
h.set_weights(np.full(P, 1.0 / P)); h.set_beams(bx, by, bz, bv); h.synchronize()
print(f"create + broadcast {P} maps of 2 MiB ({P * 2 / 1024:.1f} GiB): {time.time() - t0:.2f} s", flush=True)
h.profile(True)
for add in (1, 0):
    t0 = time.time(); h.measure(add); h.synchronize()
    print(f"measure add_points={add}: {1e3 * (time.time() - t0):.2f} ms", flush=True)
    if add:
        s, c = h.get_scores()
pick = np.arange(0, P, max(1, P // 40))
maps = np.tile(sc["grid"].reshape(1, -1), (len(pick), 1))
so, co = O.measure(O.ref_grid(*dims), sc["x"][pick], sc["y"][pick], sc["z"][pick], sc["yaw"][pick],
                   maps.reshape(-1), maps.shape[1], bx, by, bz, bv, 1)
assert np.array_equal(s[pick], so) and np.array_equal(c[pick], co), "scores differ from the oracle"
for k, i in enumerate(pick[:8]):
    assert np.array_equal(h.read_map(int(i)), maps[k]), f"map {i} differs from the oracle"
h.weights()
u = capi.draw_uniform(4, P)
t0 = time.time(); slots, best, nohit, amb = h.resample(u)
print(f"resample (moves {int((slots != np.arange(P)).sum())} map slabs of 2 MiB): {1e3 * (time.time() - t0):.2f} ms", flush=True)
for i in np.flatnonzero(slots != np.arange(P))[:4]:
    assert np.array_equal(h.read_map(int(i)), h.read_map(int(slots[i])))
for name, kid in (("measure", capi.PF_K_MEASURE), ("weights", capi.PF_K_WEIGHTS), ("resample", capi.PF_K_RESAMPLE)):
    ms, n = h.kernel_time(kid)
    print(f"  {name}: {ms / max(n, 1):.3f} ms avg over {n}")
# K2's roofline entry on this (sparse) scan: 2 B per evaluation, three more fused launches
h.profile(False); h.profile(True, [capi.PF_K_MEASURE])
for _ in range(3):
    h.measure(1)
h.synchronize()
ms, n = h.kernel_time(capi.PF_K_MEASURE)
t = ms / n * 1e-3
print(f"K2 roofline (configs[3], sparse synthetic scan): {P} particles x {len(bx)} beams: {t * 1e3:.3f} ms per launch, "
      f"algorithmic {2.0 * P * len(bx) / t / 1e9:.1f} GB/s = {2.0 * P * len(bx) / t / 8e12 * 100:.2f} % of 8 TB/s "
      f"(every beam in a 64-byte line of its own in a private 2 MiB map; the reference's dense list: tools/c4_dense.py)")
print("c4 check ok")
h.close()
