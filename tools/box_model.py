#!/usr/bin/env python3
"""Development aid: prices LDS staging for K1 on the bench scene. For (sorted particle block x
beam cluster) pairs: the union box of the cells the block's particles touch, its bytes per
evaluation, and the distinct LDS dwords / worst bank multiplicity of a wave-wide byte gather."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
G = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sc = synth.scene((G, G, G), P, M, seed=0)
bx, by, bz, bv = sc["beams"]
res = 0.05
x, y, z, yaw = sc["x"], sc["y"], sc["z"], sc["yaw"]
tx, ty, tz = x / res, y / res, z / res

def spread3(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6) | ((v & 16) << 8)
a = np.floor(yaw * 200).astype(int) & 15; kx = np.floor(tx * 3).astype(int) & 15; ky = np.floor(ty * 3).astype(int) & 15
key = (spread3(a) << 2) | (spread3(kx) << 1) | spread3(ky)
order = np.argsort(key, kind="stable")

def kd(idx, target):
    if len(idx) <= target:
        return [idx]
    pts = np.stack([bx[idx], by[idx], bz[idx]], 1)
    ax = np.argmax(pts.max(0) - pts.min(0))
    o = idx[np.argsort(pts[:, ax], kind="stable")]
    h = len(o) // 2
    return kd(o[:h], target) + kd(o[h:], target)

for target in (48, 96, 192):
    clusters = kd(np.arange(len(bx)), target)
    for WG in (256, 512, 1024):
        tot_bytes = tot_evals = 0; worst = 0; dd = []; bank = []
        rng = np.random.default_rng(0)
        nblk = (P + WG - 1) // WG
        for blk in rng.choice(nblk, min(nblk, 6), replace=False):
            pp = order[blk * WG:(blk + 1) * WG]
            c, s = np.cos(yaw[pp]), np.sin(yaw[pp])
            for ci in rng.choice(len(clusters), 12, replace=False):
                cl = clusters[ci]
                X = np.rint(c[:, None] * bx[cl][None] - s[:, None] * by[cl][None] + tx[pp][:, None]).astype(int)
                Y = np.rint(s[:, None] * bx[cl][None] + c[:, None] * by[cl][None] + ty[pp][:, None]).astype(int)
                Z = (bz[cl][None] + np.rint(tz[pp])[:, None]).astype(int)
                nx = X.max() - X.min() + 1; ny = Y.max() - Y.min() + 1
                z0 = Z.min() // 4 * 4; nz = (Z.max() - z0) // 4 * 4 + 4
                b = nx * ny * nz
                tot_bytes += b; tot_evals += X.size; worst = max(worst, b)
                # wave-wide gather: lanes = 64 consecutive particles, one beam
                for w0 in range(0, len(pp), 64):
                    for j in range(0, len(cl), max(1, len(cl) // 4)):
                        addr = ((X[w0:w0 + 64, j] - X.min()) * ny + (Y[w0:w0 + 64, j] - Y.min())) * nz + (Z[w0:w0 + 64, j] - z0)
                        dw = np.unique(addr // 4)
                        dd.append(len(dw)); bank.append(np.bincount(dw % 64).max())
        print(f"cluster<= {target:3d} ({len(clusters)} clusters) WG {WG:4d}: box avg {tot_bytes / 72 / 1024:6.1f} KB worst {worst / 1024:6.1f} KB, "
              f"{tot_bytes / tot_evals:5.2f} B staged per eval; gather: {np.mean(dd):.1f} distinct dwords, worst bank x{np.mean(bank):.2f} (max {max(bank)})")
