#!/usr/bin/env python3
"""Development aid: random scenes through the column kernels of per-particle maps (pf_fuse_cols.hip, with and
without the add) and through K1's run groups on the shared grid, every score and map byte against the
oracle. SLAM_PF_FUSE_COLS=1 is set so sparse lists take the column form too.
    python tools/fuzz_fuse_cols.py [cases] [seed]"""
import os, sys
os.environ.setdefault("SLAM_PF_FUSE_COLS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi
from oracle import binding as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def beam_list(rng, half, zlo, zhi, style):
    bx, by, bz, bv = [], [], [], []
    for lx in range(-half, half + 1):
        for ly in range(-half, half + 1):
            if style == "sparse" and rng.random() < 0.6:
                continue
            lo = zlo + int(rng.integers(0, 8)); hi = zhi - int(rng.integers(0, 8))
            hole = {"dense": 0.05, "holes": 0.4, "sparse": 0.85, "walls": 0.0}[style]
            for lz in range(lo, hi + 1):
                if rng.random() < hole:
                    continue
                v = 0
                while v == 0:
                    v = int(rng.integers(-128, 128))
                bx.append(lx); by.append(ly); bz.append(lz); bv.append(v)
    b = [np.array(a, np.int32) for a in (bx, by, bz)] + [np.array(bv, np.int8)]
    if rng.random() < 0.3 and len(bx) > 10:                     # repeats and disorder
        extra = rng.choice(len(bx), max(1, len(bx) // 15), replace=False)
        b = [np.concatenate([a, a[extra]]) for a in b]
        if rng.random() < 0.5:
            perm = rng.permutation(len(b[0]))
            b = [a[perm] for a in b]
    return tuple(b)


bad = 0
only = [int(v) for v in os.environ.get('FUZZ_ONLY', '').split(',') if v]
for case in range(cases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 1000 + case)
    gz = int(rng.choice([16, 32, 48, 80, 96, 112]))
    gx, gy = int(rng.integers(20, 56)), int(rng.integers(20, 56))
    dims = (gx, gy, gz)
    style = str(rng.choice(["dense", "holes", "sparse", "walls"]))
    half = int(rng.integers(2, 8))
    zspan = int(rng.integers(4, min(gz, 100)))
    beams = beam_list(rng, half, -zspan // 2, zspan // 2, style)
    if len(beams[0]) == 0:
        continue
    P = int(rng.integers(3, 70))
    res = 0.05
    x = rng.uniform(0.1, gx * res - 0.1, P); y = rng.uniform(0.1, gy * res - 0.1, P); z = rng.uniform(0.05, gz * res - 0.05, P)
    mid = rng.random(P) < 0.7                                    # most particles well inside
    x[mid] = rng.uniform(gx * res * 0.4, gx * res * 0.6, int(mid.sum())); y[mid] = rng.uniform(gy * res * 0.4, gy * res * 0.6, int(mid.sum()))
    z[mid] = rng.uniform(gz * res * 0.45, gz * res * 0.55, int(mid.sum()))
    tie = rng.random(P) < 0.1
    z[tie] = (np.floor(z[tie] / res) + 0.5) * res
    yaw = rng.uniform(-3.3, 3.3, P)
    thr = int(rng.choice([20, 0, -5, 60])); lo = int(rng.choice([0, 0, 7])); hi = int(rng.choice([100, 255, 90]))
    grid0 = rng.integers(0, 256, dims).astype(np.uint8)
    og = O.ref_grid(*dims, occ_threshold=thr, map_lo=lo, map_hi=hi)
    maps = np.tile(grid0.reshape(1, -1), (P, 1))
    cfg = capi.default_config(n_particles=P, grid_x=gx, grid_y=gy, grid_z=gz, local_x=0, local_y=0, local_z=0,
                              max_beams=len(beams[0]), occ_threshold=thr, map_lo=lo, map_hi=hi)
    h = capi.Handle(cfg)
    ok = True
    why = []
    try:
        h.broadcast_map(grid0); h.predict(x, y, z, yaw); h.set_beams(*beams)
        for add in (1, 0, 1):
            s, c = O.measure(og, x, y, z, yaw, maps.reshape(-1), maps.shape[1], *beams, add)
            h.measure(add)
            gs, gc = h.get_scores()
            if not (np.array_equal(gs, s) and np.array_equal(gc, c)):
                ok = False; why.append(f"scores add={add} particles {np.flatnonzero((gs != s) | (gc != c))[:6].tolist()}")
            for i in range(P):
                if not np.array_equal(h.read_map(i), maps[i]):
                    ok = False; why.append(f"map {i} add={add} cells {np.flatnonzero(h.read_map(i) != maps[i])[:4].tolist()} pose {x[i]:.4f} {y[i]:.4f} {z[i]:.4f} {yaw[i]:.4f}")
                    break
    finally:
        h.close()
    # K1 run groups on the shared grid with the same list
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=gx, grid_y=gy, grid_z=gz, local_x=0,
                              local_y=0, local_z=0, max_beams=len(beams[0]), occ_threshold=thr)
    h = capi.Handle(cfg)
    try:
        h.write_map(0, grid0); h.predict(x, y, z, yaw); h.set_beams(*beams); h.measure(0)
        gs, gc = h.get_scores()
        g1 = grid0.reshape(-1).copy()
        s, c = O.measure(O.ref_grid(*dims, occ_threshold=thr), x, y, z, yaw, g1, 0, *beams, 0)
        if not (np.array_equal(gs, s) and np.array_equal(gc, c)):
            i0 = int(np.flatnonzero((gs != s) | (gc != c))[0])
            occ = beams[3] > thr
            why.append(f"p{i0}: got ({gs[i0]}, {gc[i0]}) want ({s[i0]}, {c[i0]}); occupied {int(occ.sum())}, columns {len(set(zip(beams[0][occ].tolist(), beams[1][occ].tolist())))}, wrong {int(((gs != s) | (gc != c)).sum())} of {P}")
            ok = False; why.append(f"shared-grid K1 particles {np.flatnonzero((gs != s) | (gc != c))[:6].tolist()} z/res {(z[np.flatnonzero((gs != s) | (gc != c))[:3]] / res).tolist()}")
    finally:
        h.close()
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH dims={dims} style={style} half={half} zspan={zspan} P={P} beams={len(beams[0])} thr={thr} lo={lo} hi={hi}: " + "; ".join(why[:4]), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
