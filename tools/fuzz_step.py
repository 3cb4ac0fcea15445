#!/usr/bin/env python3
"""Development aid: random whole steps on the shared grid (measure -> update_map -> predict -> weights ->
resample, several steps in a row, the map update carried by the weight kernels) against the oracle: particle
counts on and beside every block size the kernels use, grids whose sizes are not multiples of the brick,
beam lists from one beam to thousands, weights with zeros, one-hot and tiny values, thresholds that repeat.
Scores, resampled slots, best index and the grid are compared exactly; weights exactly in the exact order
(up to 2048 particles) and to 1e-12 beyond (the fixed-shape sums differ from the sequential ones by rounding).
    python tools/fuzz_step.py [cases] [seed]        (FUZZ_RES=1: voxel sizes 0.025 to 0.1 m and occupancy thresholds 0 to 99 as well)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
from oracle import binding as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1025, 2047, 2048, 2049, 4095, 4097, 8191, 8193, 12000, 33000]
only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]
bad = 0
for case in range(cases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 7919 + case)
    P = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(1, 20000))
    dims = tuple(int(v) for v in rng.integers(24, 150, 3))
    if rng.random() < 0.4:
        dims = tuple(int(v) // 4 * 4 for v in dims)
    M = int(rng.choice([1, 2, 3, 5, 31, 33, 100, 500, 2000, 5000]))
    res = float(rng.choice([0.05, 0.05, 0.1, 0.04, 0.025])) if os.environ.get("FUZZ_RES") else 0.05
    thr = int(rng.choice([20, 20, 0, 49, 99])) if os.environ.get("FUZZ_RES") else 20
    sc = synth.scene(dims, P, M, resolution=res, seed=case, max_range_vox=int(0.6 * max(dims)))
    beams_all = sc["beams"]
    if len(beams_all[0]) == 0:
        continue
    bv = beams_all[3].copy()
    bv[rng.random(len(bv)) < 0.3] = 10                           # not occupied (S3P:886)
    beams_all = (beams_all[0], beams_all[1], beams_all[2], bv)
    kind = str(rng.choice(["uniform", "random", "zeros", "onehot", "tiny"]))
    w = {"uniform": np.full(P, 1.0 / P), "random": rng.random(P) + 1e-3, "zeros": np.where(rng.random(P) < 0.5, 0.0, rng.random(P)),
         "onehot": np.eye(1, P, int(rng.integers(0, P)))[0], "tiny": rng.random(P) * 1e-300}[kind]
    if w.sum() > 0 and kind != "tiny":
        w = w / w.sum()
    grid = sc["grid"].copy()
    x, y, z, yaw = sc["x"].copy(), sc["y"].copy(), sc["z"].copy(), sc["yaw"].copy()
    og = O.ref_grid(*dims, resolution=res, occ_threshold=thr)
    prm = (10, 100, -3, 0)
    ucfg = O.UpdateCfg(dims[0], dims[1], dims[2], res, 0, prm[0], prm[1], prm[2], prm[3], 3.0)
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                              local_x=0, local_y=0, local_z=0, max_beams=len(beams_all[0]), max_points=max(len(sc["points"]), 1),
                              resolution=res, occ_threshold=thr)
    h = capi.Handle(cfg)
    why = []
    best = 0
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_weights(w)
        for step in range(3):
            keep = np.sort(rng.choice(len(beams_all[0]), max(1, int(len(beams_all[0]) * rng.uniform(0.3, 1.0))), replace=False))
            beams = tuple(a[keep] for a in beams_all)
            h.set_beams(*beams); h.measure(0)
            s, c = O.measure(og, x, y, z, yaw, grid.reshape(-1), 0, *beams, 0)
            gs, gc = h.get_scores()
            if not (np.array_equal(gs, s) and np.array_equal(gc, c)):
                why.append(f"step {step} scores (res {res} thr {thr}): {int(((gs != s) | (gc != c)).sum())} particles"); break
            if len(sc["points"]) and rng.random() < 0.8:
                pos = sc["true_pose"][:3] + rng.normal(0, 0.05, 3); zyx = np.array([sc["true_pose"][3] + rng.normal(0, 0.05), 0.01, -0.02])
                pts = sc["points"]; ridx = np.arange(len(pts), dtype=np.int32)
                h.set_scan(pts, ridx); h.update_map(capi.PF_GRID_SHARED, capi.UpdateParams(*prm, 3.0), pos, zyx)
                grid_before = grid.copy(); upd_in = (pos.copy(), zyx.copy())
                O.update_map(ucfg, grid, pos, zyx, pts, ridx)
            x = x + rng.normal(0, 0.02, P); y = y + rng.normal(0, 0.02, P); yaw = yaw + rng.normal(0, 0.01, P)
            h.predict(x, y, z, yaw)
            w_pre = w.copy()
            O.weights(w_pre, s, int(c[-1]))
            h.weights()
            gw, _ = h.get_weights()
            exact = P <= 2048
            okw = np.array_equal(gw, w_pre, equal_nan=True) if exact else np.allclose(gw, w_pre, rtol=1e-12, atol=0, equal_nan=True)
            if not okw:
                why.append(f"step {step} weights ({kind})"); break
            u = capi.draw_uniform(1000 * case + step, P) if rng.random() < 0.8 else np.full(P, float(rng.random()))
            # the oracle resamples the weights the GPU holds (beyond 2048 particles they differ from the
            # sequential sums in the last place, and a slot may legitimately follow its own weights)
            w_res = gw.copy()
            d, sa, obest, nohit = O.resample(w_res, u, best)
            slots, gbest, gnohit, amb = h.resample(u)
            if not (np.array_equal(slots, sa) and gbest == obest and gnohit == nohit):
                why.append(f"step {step} resample ({kind}): {int((slots != sa).sum())} slots, best {gbest}/{obest}, nohit {gnohit}/{nohit}, amb {amb}"); break
            best = obest
            x, y, z, yaw = x[sa], y[sa], z[sa], yaw[sa]
            w = w_res
            gw2, _ = h.get_weights()
            if not (np.array_equal(gw2, w, equal_nan=True) if exact else np.allclose(gw2, w, rtol=1e-12, atol=0, equal_nan=True)):
                why.append(f"step {step} weights after resample"); break
            w = gw2
            px, py, pz, pyaw = h.get_poses()
            if not (np.array_equal(px, x) and np.array_equal(pyaw, yaw)):
                why.append(f"step {step} poses after resample"); break
            got = h.read_map(0).reshape(dims)
            if not np.array_equal(got, grid):
                idx = np.argwhere(got != grid)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_step_case{case}.npz"), before=grid_before, got=got, want=grid,
                                    pos=upd_in[0], zyx=upd_in[1], pts=sc["points"], dims=np.array(dims))
                why.append(f"step {step} grid after update: {len(idx)} cells, first {idx[:3].tolist()} got {[int(got[tuple(i)]) for i in idx[:3]]} want {[int(grid[tuple(i)]) for i in idx[:3]]}"); break
    finally:
        h.close()
    if why:
        bad += 1
        print(f"case {case}: MISMATCH P={P} dims={dims} M={len(beams_all[0])} weights={kind}: " + "; ".join(why), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
