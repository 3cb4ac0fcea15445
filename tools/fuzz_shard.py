#!/usr/bin/env python3
"""Development aid: every rank's kernels of a `world`-rank job, run one after the other on this one GPU, against
the one-GPU handle on the same particles (which tools/fuzz_step.py checks against the oracle): random world
sizes and shard sizes on and beside the block sizes, weights with zeros, one-hot and tiny values, thresholds that
repeat, the map update riding in the weight kernels on every rank, two steps in a row so that the resampled
state, the best index and the weights carry over. Scores and pose blocks are gathered by hand in rank order, as
the all-gathers would. Everything is compared bit for bit.
    python tools/fuzz_shard.py [cases] [seed]"""
import os, sys
import numpy as np
import torch                                                       # before the library: one HIP runtime per process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]
SHARD = [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2049, 4097, 6000]
bad = 0
for case in range(cases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 32452843 + case)
    world = int(rng.choice([2, 3, 4, 8]))
    P = int(rng.choice(SHARD)) if rng.random() < 0.7 else int(rng.integers(1, 5000))
    Pg = P * world
    dims = tuple(int(v) for v in rng.integers(24, 110, 3))
    M = int(rng.choice([1, 5, 33, 100, 700, 3000]))
    sc = synth.scene(dims, Pg, M, seed=case, max_range_vox=int(0.6 * max(dims)))
    if len(sc["beams"][0]) == 0:
        continue
    kind = str(rng.choice(["uniform", "random", "zeros", "onehot", "tiny"]))
    w0 = {"uniform": np.full(Pg, 1.0 / Pg), "random": rng.random(Pg) + 1e-3, "zeros": np.where(rng.random(Pg) < 0.5, 0.0, rng.random(Pg)),
          "onehot": np.eye(1, Pg, int(rng.integers(0, Pg)))[0], "tiny": rng.random(Pg) * 1e-300}[kind]
    if w0.sum() > 0 and kind != "tiny":
        w0 = w0 / w0.sum()
    order = capi.PF_ORDER_FAST if (Pg > 2048 or rng.random() < 0.5) else capi.PF_ORDER_EXACT
    def mk(n, **kw):
        return capi.Handle(capi.default_config(n_particles=n, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                                               local_x=0, local_y=0, local_z=0, max_beams=len(sc["beams"][0]), max_points=max(len(sc["points"]), 1),
                                               order_mode=order, **kw))
    full = mk(Pg)
    shards = [mk(P, n_particles_global=Pg, particle_offset=r * P) for r in range(world)]
    why = []
    try:
        x, y, z, yaw = (sc[k].copy() for k in ("x", "y", "z", "yaw"))
        for hh, lo in [(full, None)] + [(s, r * P) for r, s in enumerate(shards)]:
            hh.write_map(0, sc["grid"]); hh.set_weights(w0)
            sl = slice(None) if lo is None else slice(lo, lo + P)
            hh.predict(x[sl], y[sl], z[sl], yaw[sl])
        prm = (10, 100, -3, 0)
        for step in range(2):
            keep = np.sort(rng.choice(len(sc["beams"][0]), max(1, int(len(sc["beams"][0]) * rng.uniform(0.3, 1.0))), replace=False))
            beams = tuple(a[keep] for a in sc["beams"])
            upd = len(sc["points"]) > 0 and rng.random() < 0.7
            pos = sc["true_pose"][:3] + rng.normal(0, 0.05, 3); zyx = np.array([sc["true_pose"][3] + rng.normal(0, 0.05), 0.01, -0.02])
            u = capi.draw_uniform(1000 * case + step, Pg) if rng.random() < 0.8 else np.full(Pg, float(rng.random()))
            dx, dy, dyaw = rng.normal(0, 0.02, Pg), rng.normal(0, 0.02, Pg), rng.normal(0, 0.01, Pg)
            def front(hh):
                hh.set_beams(*beams); hh.measure(0)
                if upd:
                    hh.set_scan(sc["points"], np.arange(len(sc["points"]), dtype=np.int32))
                    hh.update_map(capi.PF_GRID_SHARED, capi.UpdateParams(*prm, 3.0), pos, zyx)
            front(full)
            fs, fc = full.get_scores()
            fx, fy, fz, fyaw = full.get_poses()
            full.predict(fx + dx, fy + dy, fz, fyaw + dyaw)
            full.synchronize()
            pose_full = full.poses_tensor().clone()
            full.weights()
            fw, flw = full.get_weights()
            fslots, fbest, fnohit, famb = full.resample(u)
            fw2, flw2 = full.get_weights()
            fposes = np.stack(full.get_poses())
            fgrid = full.read_map(0)
            sc_parts = []
            for r, hh in enumerate(shards):
                lo = r * P
                front(hh)
                s, c = hh.get_scores()
                if not (np.array_equal(s, fs[lo:lo + P]) and np.array_equal(c, fc[lo:lo + P])):
                    why.append(f"step {step} rank {r} scores"); break
                sc_parts.append(np.stack([s, c], 1).astype(np.int32))
                px, py, pz, pyaw = hh.get_poses()
                hh.predict(px + dx[lo:lo + P], py + dy[lo:lo + P], pz, pyaw + dyaw[lo:lo + P])
            if why:
                break
            scores_all = torch.from_numpy(np.concatenate(sc_parts, 0)).cuda().contiguous()
            poses_all = torch.cat([pose_full[:, r * P:(r + 1) * P].contiguous().reshape(-1) for r in range(world)]).contiguous()
            torch.cuda.synchronize()
            for r, hh in enumerate(shards):
                lo = r * P
                hh.synchronize()
                if not torch.equal(hh.poses_tensor(), pose_full[:, lo:lo + P]):
                    why.append(f"step {step} rank {r} pose block after predict"); break
                hh.weights(scores_all.data_ptr())
                w, lw = hh.get_weights()
                if not (np.array_equal(w, fw, equal_nan=True) and np.array_equal(lw, flw, equal_nan=True)):
                    why.append(f"step {step} rank {r} weights ({kind}): {int((w != fw).sum())} differ"); break
                slots, best, nohit, amb = hh.resample(u, poses_all.data_ptr())
                if not (np.array_equal(slots, fslots) and (best, nohit, amb) == (fbest, fnohit, famb)):
                    why.append(f"step {step} rank {r} resample ({kind}): {int((slots != fslots).sum())} slots, best {best}/{fbest} nohit {nohit}/{fnohit} amb {amb}/{famb}"); break
                w2, lw2 = hh.get_weights()
                if not (np.array_equal(w2, fw2, equal_nan=True) and np.array_equal(lw2, flw2, equal_nan=True)):
                    why.append(f"step {step} rank {r} weights after resample"); break
                if not np.array_equal(np.stack(hh.get_poses()), fposes[:, lo:lo + P]):
                    why.append(f"step {step} rank {r} poses after resample"); break
                if not np.array_equal(hh.read_map(0), fgrid):
                    why.append(f"step {step} rank {r} grid after the carried update"); break
            if why:
                break
    except capi.PfError as e:
        why.append(f"PfError: {e}")
    finally:
        full.close()
        for s in shards:
            s.close()
    if why:
        bad += 1
        print(f"case {case}: MISMATCH world={world} P={P} dims={dims} M={len(sc['beams'][0])} weights={kind} order={order}: " + "; ".join(why), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
