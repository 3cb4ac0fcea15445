#!/usr/bin/env python3
"""Development aid (host only): distinct 64-B / 128-B lines per wave-wide gather of K1 for
candidate grid layouts and particle orders, on the bench scene."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from slam_amd import synth

P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
bx, by, bz, _ = sc["beams"]
x, y, z, yaw = sc["x"] / 0.05, sc["y"] / 0.05, sc["z"] / 0.05, sc["yaw"]
sub = np.arange(0, len(bx), 16)
lx, ly, lz = bx[sub].astype(float), by[sub].astype(float), bz[sub].astype(float)

def spread3(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6) | ((v & 16) << 8)

def key(qa, qx, qy, mode):
    a = np.floor(yaw * qa).astype(int); xx = np.floor(x * qx).astype(int); yy = np.floor(y * qy).astype(int)
    if mode == 0:
        return ((a & 63) << 9) | ((xx & 15) << 5) | (yy & 31)
    return (spread3(a & 31) << 2) | (spread3(xx & 31) << 1) | spread3(yy & 31)

layouts = {
    "linear z-fastest      ": lambda X, Y, Z: (X * G + Y) * G + Z,
    "[Z/8][X][Y][Z%8]      ": lambda X, Y, Z: (((Z >> 3) * G + X) * G + Y) * 8 + (Z & 7),
    "[X][Y/8][Z/8] 8x8 tile": lambda X, Y, Z: ((X * (G // 8) + (Y >> 3)) * (G // 8) + (Z >> 3)) * 64 + (Y & 7) * 8 + (Z & 7),
    "4x4x4 bricks          ": lambda X, Y, Z: (((X >> 2) * (G // 4) + (Y >> 2)) * (G // 4) + (Z >> 2)) * 64 + (X & 3) * 16 + (Y & 3) * 4 + (Z & 3),
    "8x8x2 bricks (128 B)  ": lambda X, Y, Z: (((X >> 3) * (G // 8) + (Y >> 3)) * (G // 2) + (Z >> 1)) * 128 + (X & 7) * 16 + (Y & 7) * 2 + (Z & 1),
    "8x4x4 bricks (128 B)  ": lambda X, Y, Z: (((X >> 3) * (G // 4) + (Y >> 2)) * (G // 4) + (Z >> 2)) * 128 + (X & 7) * 16 + (Y & 3) * 4 + (Z & 3),
    "4x4x8 bricks (128 B)  ": lambda X, Y, Z: (((X >> 2) * (G // 4) + (Y >> 2)) * (G // 8) + (Z >> 3)) * 128 + (X & 3) * 32 + (Y & 3) * 8 + (Z & 7),
}

def stats(order, name):
    o = order[: (P // 64) * 64].reshape(-1, 64)[::4]
    c, s = np.cos(yaw), np.sin(yaw)
    res = {k: [[], []] for k in layouts}
    for w in o:
        X = np.rint(c[w, None] * lx - s[w, None] * ly + x[w, None]).astype(np.int64)
        Y = np.rint(s[w, None] * lx + c[w, None] * ly + y[w, None]).astype(np.int64)
        Z = np.rint(lz + z[w, None]).astype(np.int64)
        for k, fn in layouts.items():
            f = fn(X, Y, Z)
            s64 = np.sort(f >> 6, axis=0); s128 = np.sort(f >> 7, axis=0)
            res[k][0].append(1 + (np.diff(s64, axis=0) != 0).sum(0))
            res[k][1].append(1 + (np.diff(s128, axis=0) != 0).sum(0))
    print(name)
    for k in layouts:
        a = np.concatenate(res[k][0]); b = np.concatenate(res[k][1])
        print(f"   {k}  64-B lines {a.mean():5.2f} (p90 {np.percentile(a, 90):3.0f})   128-B lines {b.mean():5.2f}")

stats(np.argsort(key(500, 2, 4, 0), kind="stable"), "legacy key")
stats(np.argsort(key(300, 3, 3, 1), kind="stable"), "morton 300,3,3")
R = np.sqrt(np.mean(lx ** 2 + ly ** 2))
pts = np.stack([yaw * R, x, y, z], 1)
def kd(idx, dims):
    if len(idx) <= 64:
        return [idx]
    ext = pts[idx][:, :dims].max(0) - pts[idx][:, :dims].min(0)
    d = int(np.argmax(ext))
    srt = idx[np.argsort(pts[idx, d])]
    h = (len(srt) // 2 + 63) // 64 * 64
    return kd(srt[:h], dims) + kd(srt[h:], dims)
stats(np.concatenate(kd(np.arange(P), 3)), "k-d (yaw*R, x, y)")
stats(np.concatenate(kd(np.arange(P), 4)), "k-d (yaw*R, x, y, z)")
