#!/usr/bin/env python3
"""Development aid: per-wave timeline of one K1 launch (SLAM_PF_K1_TRACE dump)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth
P, M, G = 10000, 4096, 256
sc = synth.scene((G, G, G), P, M, seed=0)
cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                          local_x=0, local_y=0, local_z=0, max_beams=M)
h = capi.Handle(cfg)
h.write_map(0, sc["grid"]); h.set_beams(*sc["beams"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
for _ in range(5):
    h.measure(0)
h.synchronize()
os.environ["SLAM_PF_K1_TRACE"] = "gpurun_out/k1_trace.bin"
h.measure(0); h.synchronize()
del os.environ["SLAM_PF_K1_TRACE"]
t = np.fromfile("gpurun_out/k1_trace.bin", np.uint64).reshape(-1, 4)
full = t.copy()
t = t[t[:, 2] > 0]
t0 = t[:, 0].min()
st, lp, en = (t[:, 0] - t0).astype(float), (t[:, 1] - t0).astype(float), (t[:, 2] - t0).astype(float)
tick = 1e-2   # wall_clock64: 100 MHz -> us
print("waves", len(t), "kernel span us", en.max() * tick)
for name, v in (("start", st), ("loop start", lp), ("end", en), ("life", en - st), ("prologue", lp - st), ("loop", en - lp)):
    print(f"{name:10s} us: min {v.min() * tick:7.2f} p10 {np.percentile(v, 10) * tick:7.2f} p50 {np.percentile(v, 50) * tick:7.2f} "
          f"p90 {np.percentile(v, 90) * tick:7.2f} p99 {np.percentile(v, 99) * tick:7.2f} max {v.max() * tick:7.2f}")
hw = t[:, 3] & 0xffffffff
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("distinct (se,sh,cu)", len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))))
# how much of the spread of wave lives is explained by the particle block (x), the wave in it, the chunk (y)
nx = (P + 255) // 256
ny = len(full) // (nx * 4)
life = (full[:, 2].astype(float) - full[:, 0].astype(float)).reshape(ny, nx, 4) * tick
valid = (full[:, 2] > 0).reshape(ny, nx, 4)
life = np.where(valid, life, np.nan)
print("chunks", ny, "particle blocks", nx, "overall std %.2f us" % np.nanstd(life))
pw = np.nanmean(life, axis=0)                      # per particle wave (x, wave)
print("per particle-wave mean life: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f, std of means %.2f" % (
    np.nanmin(pw), np.nanpercentile(pw, 10), np.nanpercentile(pw, 50), np.nanpercentile(pw, 90), np.nanmax(pw), np.nanstd(pw)))
pc = np.nanmean(life, axis=(1, 2))
print("per chunk mean life: min %.2f max %.2f std %.2f" % (pc.min(), pc.max(), pc.std()))
resid = life - pw[None]
print("std after removing the particle-wave mean: %.2f us" % np.nanstd(resid))
h.close()
