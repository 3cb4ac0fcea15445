#!/usr/bin/env python3
"""Development aid (host only): dump K1's real per-wave gather address vectors on the bench scene
(legacy particle order) for tools/ubench_gather3. Output: tools/k1_addr.bin, uint32 [W][B][64]."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from slam_amd import synth
P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
bx, by, bz, _ = sc["beams"]
x, y, z, yaw = sc["x"] / 0.05, sc["y"] / 0.05, sc["z"] / 0.05, sc["yaw"]
a = np.floor(yaw * 500).astype(int); xx = np.floor(x * 2).astype(int); yy = np.floor(y * 4).astype(int)
key = ((a & 63) << 9) | ((xx & 15) << 5) | (yy & 31)
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
        f = (((Zi >> 3) * G + Xi) * G + Yi) * 8 + (Zi & 7)
    else:
        f = (Xi * G + Yi) * G + Zi
    out[w] = np.clip(f, 0, G ** 3 - 1).T.astype(np.uint32)
out.tofile("tools/k1_addr.bin")
print(out.shape, "lines/gather", np.mean([len(np.unique(v >> 6)) for v in out.reshape(-1, 64)[::37]]),
      "128B lines", np.mean([len(np.unique(v >> 7)) for v in out.reshape(-1, 64)[::37]]))
