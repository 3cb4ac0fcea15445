#!/usr/bin/env python3
"""Development aid (host only): distinct 64-B lines / dwords per wave-wide gather of K1 for a
given particle-order key, on the bench scene. Mirrors order_key in pf_kernels.hip."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from slam_amd import synth

P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
bx, by, bz, _ = sc["beams"]
x, y, z, yaw = sc["x"] / 0.05, sc["y"] / 0.05, sc["z"] / 0.05, sc["yaw"]
sub = np.arange(0, len(bx), 16)                       # a sample of the beams
lx, ly, lz = bx[sub].astype(float), by[sub].astype(float), bz[sub].astype(float)

def spread3(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6) | ((v & 16) << 8)

def key(qa, qx, qy, mode):
    a = np.floor(yaw * qa).astype(int); xx = np.floor(x * qx).astype(int); yy = np.floor(y * qy).astype(int)
    if mode == 0:
        return ((a & 63) << 9) | ((xx & 15) << 5) | (yy & 31)
    if mode == 1:
        return (spread3(a & 31) << 2) | (spread3(xx & 31) << 1) | spread3(yy & 31)
    if mode == 3:   # lexicographic rank-based: quantiles
        return None

def stats(order, name):
    o = order[: (P // 64) * 64].reshape(-1, 64)
    c, s = np.cos(yaw), np.sin(yaw)
    lines = []; dwords = []
    for w in o[::4]:
        X = np.rint(c[w, None] * lx - s[w, None] * ly + x[w, None])
        Y = np.rint(s[w, None] * lx + c[w, None] * ly + y[w, None])
        Z = np.rint(lz + z[w, None])
        f = (X * G * G + Y * G + Z).astype(np.int64)       # [64, nb]
        for j in range(f.shape[1]):
            lines.append(len(np.unique(f[:, j] >> 6))); dwords.append(len(np.unique(f[:, j] >> 2)))
    print(f"{name:28s} lines/gather mean {np.mean(lines):5.2f} p90 {np.percentile(lines, 90):4.0f}   dwords/gather mean {np.mean(dwords):5.2f}")

stats(np.arange(P), "unordered")
for k in [(500, 2, 4, 0), (200, 2, 2, 1), (100, 1, 1, 1), (300, 3, 3, 1)]:
    stats(np.argsort(key(*k), kind="stable"), f"key {k}")
# recursive median split (k-d order) on (yaw*R, x, y) with R = rms beam range
R = np.sqrt(np.mean(lx ** 2 + ly ** 2))
pts = np.stack([yaw * R, x, y], 1)
def kd(idx, depth):
    if len(idx) <= 64:
        return [idx]
    ext = pts[idx].max(0) - pts[idx].min(0)
    d = int(np.argmax(ext))
    srt = idx[np.argsort(pts[idx, d])]
    h = (len(srt) // 2 + 63) // 64 * 64
    return kd(srt[:h], depth + 1) + kd(srt[h:], depth + 1)
stats(np.concatenate(kd(np.arange(P), 0)), f"k-d median split R={R:.0f}")
