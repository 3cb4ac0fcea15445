#!/usr/bin/env python3
"""What one rank of a W-GPU job computes per step, measured on ONE GPU.

    python tools/rehearse_world.py --world 8 [--steps 200]

The handle is created as rank 0 of W (10 k local particles, W x 10 k global); the two all-gathers
are replaced by a device-side replication of the local block (every "rank" holds the same
particles), enqueued on the handle's stream, so the number is the per-rank compute of the step
without the collectives' latency: K1 on the local shard, K3/K4 redundantly over all W x 10 k
particles. The difference to bench.py's N=1 line is what the redundant stages cost as W grows;
the difference between this and the driver's N=W line is what RCCL adds.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--beams", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--fresh", type=int, default=0, help="cycle through this many pre-drawn threshold sets (0: the same thresholds every step)")
    a = ap.parse_args()
    import torch
    from slam_amd import capi, synth

    W, P, M, dims = a.world, a.particles, a.beams, (a.grid,) * 3
    Pg = P * W
    sc = synth.scene(dims, P, M, seed=0)
    rng = np.random.default_rng(1000)
    tp = sc["true_pose"]
    x = tp[0] + rng.normal(0, 0.03, P); y = tp[1] + rng.normal(0, 0.03, P)
    z = tp[2] + rng.normal(0, 0.015, P); yaw = tp[3] + rng.normal(0, 0.02, P)
    M = len(sc["beams"][0])
    cfg = capi.default_config(device=0, n_particles=P, n_particles_global=Pg, particle_offset=0,
                              map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                              local_x=0, local_y=0, local_z=0, max_beams=max(M, 1),
                              max_points=max(len(sc["points"]), 1))
    h = capi.Handle(cfg)
    st = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(st)
    h.set_stream(st.cuda_stream)
    h.write_map(0, sc["grid"])
    h.predict(x, y, z, yaw)
    h.set_weights(np.full(Pg, 1.0 / Pg))
    h.set_beams(*sc["beams"])
    h.set_scan(sc["points"], sc["ray_idx"])
    u = capi.draw_uniform(3, Pg)
    upd = capi.UpdateParams(10, 100, -3, 0, 3.0)
    pos = tp[:3].copy(); zyx = np.array([tp[3], 0.0, 0.0])
    NBLK = 8
    blocks = torch.empty((NBLK, 9, P), dtype=torch.float64, device="cuda:0")
    for b in range(NBLK):
        r2 = np.random.default_rng(5000 + b)
        h.predict(tp[0] + r2.normal(0, 0.03, P), tp[1] + r2.normal(0, 0.03, P),
                  tp[2] + r2.normal(0, 0.015, P), tp[3] + r2.normal(0, 0.02, P))
        h.synchronize()
        blocks[b].copy_(h.poses_tensor())
    torch.cuda.synchronize()
    scores = h.scores_tensor()
    scores_all = torch.empty((W * scores.shape[0], scores.shape[1]), dtype=scores.dtype, device="cuda:0")
    poses_all = torch.empty((W * 9, P), dtype=torch.float64, device="cuda:0")
    n = [0]

    def enqueue():
        h.measure(0)
        h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)
        blk = blocks[n[0] % NBLK]
        n[0] += 1
        h.predict_device(blk.data_ptr())
        scores_all.view(W, -1, 2).copy_(scores.unsqueeze(0).expand(W, -1, -1))     # stands in for C1
        h.weights(scores_all.data_ptr())
        poses_all.view(W, 9, P).copy_(blk.unsqueeze(0).expand(W, -1, -1))          # stands in for C2
        return poses_all.data_ptr()

    h.resample(u, enqueue(), False)
    for _ in range(a.warmup):
        h.resample(None, enqueue(), False)
    h.synchronize(); torch.cuda.synchronize()
    sets = [capi.draw_uniform(900 + k, Pg) for k in range(a.fresh)]
    amb_steps = amb_total = 0
    t0 = time.perf_counter()
    for i in range(a.steps):
        h.resample_async(sets[i % a.fresh] if a.fresh else None, enqueue(), False)
        if i:
            r = h.resample_result(); amb_steps += r[-1] > 0; amb_total += r[-1]
    r = h.resample_result(); amb_steps += r[-1] > 0; amb_total += r[-1]
    h.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"world_rehearsed": W, "particles_local": P, "particles_global": Pg, "beams": M,
                      "ms_per_step": el / a.steps * 1e3,
                      "evals_per_s_per_rank": P * M * a.steps / el,
                      "steps": a.steps, "steps_with_ambiguous_draws": int(amb_steps), "ambiguous_draws": int(amb_total),
                      "note": "one rank's compute, collectives replaced by device copies"}), flush=True)
    h.close()


if __name__ == "__main__":
    main()
