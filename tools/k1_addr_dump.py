#!/usr/bin/env python3
"""Development aid (host only): dump K1's real per-wave gather address vectors on the bench scene
(the default Morton particle order) for tools/ubench_gather3. Output: tools/k1_addr.bin, uint32 [W][B][64]."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from slam_amd import synth
P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
bx, by, bz, _ = sc["beams"]
x, y, z, yaw = sc["x"] / 0.05, sc["y"] / 0.05, sc["z"] / 0.05, sc["yaw"]
def spread3(v):
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)
a = np.floor(yaw * 200).astype(int); xx = np.floor(x * 3).astype(int); yy = np.floor(y * 3).astype(int)
key = (spread3(a & 15) << 2) | (spread3(xx & 15) << 1) | spread3(yy & 15)
order = np.argsort(key, kind="stable")[:156 * 64].reshape(156, 64)
B = 512
lx, ly, lz = bx[:B].astype(float), by[:B].astype(float), bz[:B].astype(float)
out = np.empty((156, B, 64), np.uint32)
c, s = np.cos(yaw), np.sin(yaw)
for w, idx in enumerate(order):
    X = np.rint(c[idx, None] * lx - s[idx, None] * ly + x[idx, None])
    Y = np.rint(s[idx, None] * lx + c[idx, None] * ly + y[idx, None])
    Z = np.rint(lz + z[idx, None])
    Xi, Yi, Zi = X.astype(np.int64), Y.astype(np.int64), Z.astype(np.int64)
    if len(sys.argv) > 1 and sys.argv[1] == "zoct":
        n4 = (G + 3) // 4                                     # 4x4x4 bricks, as swz_addr in pf_kernels.hip
        f = (((Xi >> 2) * n4 + (Yi >> 2)) * n4 + (Zi >> 2)) * 64 + (Xi & 3) * 16 + (Yi & 3) * 4 + (Zi & 3)
    elif len(sys.argv) > 1 and sys.argv[1] == "brick":
        f = ((((Xi >> 2) * (G // 4) + (Yi >> 2)) * (G // 4) + (Zi >> 2)) * 64 + (Xi & 3) * 16 + (Yi & 3) * 4 + (Zi & 3))
    else:
        f = (Xi * G + Yi) * G + Zi
    out[w] = np.clip(f, 0, G ** 3 - 1).T.astype(np.uint32)
out.tofile("tools/k1_addr.bin")
print(out.shape, "lines/gather", np.mean([len(np.unique(v >> 6)) for v in out.reshape(-1, 64)[::37]]),
      "128B lines", np.mean([len(np.unique(v >> 7)) for v in out.reshape(-1, 64)[::37]]))
