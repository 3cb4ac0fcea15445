#!/usr/bin/env python3
"""Development aid: random whole steps on per-particle maps (FastSLAM mode: measure with or without the
clamp-add fuse -> predict -> weights -> resample with the slabs following their ancestors), three steps in a
row, against the oracle: every score, weight, slot, pose and every byte of every particle's map after every
step. The library picks its own kernels (column form on dense lists, wave per particle otherwise).
    python tools/fuzz_pp.py [cases] [seed]"""
import os, sys
import numpy as np
import torch                                                       # before the library: one HIP runtime per process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi
from oracle import binding as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]
bad = 0
seen = {}
for case in range(cases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 15485863 + case)
    P = int(rng.choice([1, 2, 3, 50, 63, 64, 65, 100, 255, 257, 400]))
    dims = tuple(int(v) for v in rng.integers(12, 56, 3))
    if rng.random() < 0.4:
        dims = tuple(max(12, v // 4 * 4) for v in dims)
    ext = [int(rng.integers(2, max(3, d // 2))) for d in dims]      # beam box half-extents
    box = [2 * e + 1 for e in ext]
    nbox = box[0] * box[1] * box[2]
    dens = float(rng.choice([0.002, 0.05, 0.4, 1.0]))
    M = max(1, int(nbox * dens))
    lin = np.sort(rng.choice(nbox, M, replace=False))
    bx = (lin // (box[1] * box[2]) - ext[0]).astype(np.int32)
    by = ((lin // box[2]) % box[1] - ext[1]).astype(np.int32)
    bz = (lin % box[2] - ext[2]).astype(np.int32)
    bv = rng.choice(np.array([-50, -9, -3, 3, 10, 19, 20, 21, 30, 50, 127, -128], np.int8), M)
    if rng.random() < 0.2:                                         # disorder and repeats
        k = rng.integers(0, M, M); bx, by, bz, bv = bx[k], by[k], bz[k], bv[k]
    grid0 = rng.integers(0, 101, dims).astype(np.uint8)
    if rng.random() < 0.3:
        grid0[:] = rng.choice([0, 100, 50])
    res = 0.05
    ctr = np.array(dims) * res / 2
    x = ctr[0] + rng.normal(0, 0.2, P); y = ctr[1] + rng.normal(0, 0.2, P); z = ctr[2] + rng.normal(0, 0.1, P)
    yaw = rng.uniform(-3.2, 3.2, P)
    if rng.random() < 0.3:
        yaw = rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi, 0.40005245828215963], P)
        x = np.round(x / 0.025) * 0.025; y = np.round(y / 0.025) * 0.025; z = np.round(z / 0.025) * 0.025
    kind = str(rng.choice(["uniform", "random", "zeros", "onehot"]))
    w = {"uniform": np.full(P, 1.0 / P), "random": rng.random(P) + 1e-3, "zeros": np.where(rng.random(P) < 0.5, 0.0, rng.random(P)),
         "onehot": np.eye(1, P, int(rng.integers(0, P)))[0]}[kind]
    if w.sum() > 0:
        w = w / w.sum()
    og = O.ref_grid(*dims)
    maps = np.tile(grid0.reshape(1, -1), (P, 1))
    cfg = capi.default_config(n_particles=P, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2], local_x=0, local_y=0, local_z=0,
                              max_beams=M, max_points=16)
    h = capi.Handle(cfg)
    why = []
    best = 0
    try:
        h.broadcast_map(grid0); h.predict(x, y, z, yaw); h.set_weights(w)
        for step in range(3):
            if step == 0 or rng.random() < 0.5:
                keep = np.sort(rng.choice(M, max(1, int(M * rng.uniform(0.3, 1.0))), replace=False))
                beams = (bx[keep], by[keep], bz[keep], bv[keep])
                h.set_beams(*beams)
            add = int(rng.random() < 0.7)
            s, c = O.measure(og, x, y, z, yaw, maps.reshape(-1), maps.shape[1], *beams, add)
            h.measure(add)
            gs, gc = h.get_scores()
            seen["steps"] = seen.get("steps", 0) + 1; seen["fuses"] = seen.get("fuses", 0) + add
            if not (np.array_equal(gs, s) and np.array_equal(gc, c)):
                why.append(f"step {step} add={add} scores: {int(((gs != s) | (gc != c)).sum())} particles"); break
            bad_maps = [i for i in range(P) if not np.array_equal(h.read_map(i), maps[i])] if P <= 100 or step == 2 else []
            if bad_maps:
                why.append(f"step {step} add={add} maps after measure: {len(bad_maps)} particles, first {bad_maps[:3]}"); break
            x = x + rng.normal(0, 0.02, P); y = y + rng.normal(0, 0.02, P); yaw = yaw + rng.normal(0, 0.01, P)
            h.predict(x, y, z, yaw)
            O.weights(w, s, int(c[-1])); h.weights()
            gw, _ = h.get_weights()
            if not np.array_equal(gw, w, equal_nan=True):
                why.append(f"step {step} weights ({kind})"); break
            u = capi.draw_uniform(1000 * case + step, P) if rng.random() < 0.8 else np.full(P, float(rng.random()))
            d, sa, obest, nohit = O.resample(w, u, best)
            slots, gbest, gnohit, amb = h.resample(u)
            if not (np.array_equal(slots, sa) and gbest == obest and gnohit == nohit):
                why.append(f"step {step} resample ({kind}): {int((slots != sa).sum())} slots, best {gbest}/{obest}, nohit {gnohit}/{nohit}"); break
            best = obest
            seen["slots moved"] = seen.get("slots moved", 0) + int((sa != np.arange(P)).sum())
            O.permute(sa, maps)
            x, y, z, yaw = x[sa], y[sa], z[sa], yaw[sa]
            px, py, pz, pyaw = h.get_poses()
            if not (np.array_equal(px, x) and np.array_equal(pyaw, yaw)):
                why.append(f"step {step} poses after resample"); break
            bad_maps = [i for i in range(P) if not np.array_equal(h.read_map(i), maps[i])]
            if bad_maps:
                why.append(f"step {step} maps after resample: {len(bad_maps)} particles, first {bad_maps[:3]}"); break
    except capi.PfError as e:
        why.append(f"PfError: {e}")
    finally:
        h.close()
    if why:
        bad += 1
        print(f"case {case}: MISMATCH P={P} dims={dims} M={M} dens={dens} weights={kind}: " + "; ".join(why), flush=True)
print("exercised:", ", ".join(f"{k} {v}" for k, v in seen.items()), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
