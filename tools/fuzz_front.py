#!/usr/bin/env python3
"""Development aid: random cases for the callers either side of the filter step, against the oracle: the scan
vote (K6, with its lattice parameters varied, and the entry point that allocates nothing), the keyframe ring
over a dozen frames (slot replacement, recentring, painting), the depth front-end at random image shapes and
pool sizes, the map read-out at random strides and thresholds, and the map update on the int8 local map. Many
coordinates sit on whole numbers of half voxels, so that round() is on ties which the last place decides.
Everything is compared exactly.
    python tools/fuzz_front.py [cases] [seed]"""
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
seen = {}                                                          # what the cases exercised


def note(k, v=1):
    seen[k] = seen.get(k, 0) + int(v)


def snap(rng, a, frac=0.5):
    """some coordinates onto whole numbers of quarter voxels (0.0125 m)"""
    a = np.array(a, np.float64)
    m = rng.random(a.shape) < frac
    a[m] = np.round(a[m] / 0.0125) * 0.0125
    return a


def cloud_for(rng, grid_occ, dims, pos, zyx, n, res=0.05):
    R = np.array(O.rot_zyx(*zyx)).reshape(3, 3)
    if len(grid_occ) and n:
        wp = grid_occ[rng.integers(0, len(grid_occ), n)] * res + rng.normal(0, 0.02, (n, 3))
    else:
        wp = rng.uniform(0, 1, (n, 3)) * np.array(dims) * res
    pts = (wp - pos) @ R
    if n > 8:
        pts[:3] *= 40.0
        pts[3] = [1e5, -1e5, 3e5]
    return snap(rng, pts)


def shared(dims, max_points, res=0.05):
    return capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                                           grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=16, max_points=max(max_points, 1),
                                           resolution=res))


def pick_res(rng):
    """FUZZ_RES=1: voxel sizes other than the reference's 0.05 m as well"""
    return float(rng.choice([0.05, 0.1, 0.04, 0.025])) if os.environ.get("FUZZ_RES") else 0.05


def one_vote(rng, why):
    dims = tuple(int(v) for v in rng.integers(20, 140, 3))
    grid = (rng.random(dims) < rng.choice([0.02, 0.2, 0.6])).astype(np.uint8)
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 255, 257, 500, 1000, 4096]))
    res = pick_res(rng)
    pos = snap(rng, np.array(dims) * res * rng.uniform(0.2, 0.8, 3))
    zyx = np.array([rng.uniform(-3.2, 3.2), rng.normal(0, 0.03), rng.normal(0, 0.03)])
    if rng.random() < 0.3:
        zyx[:] = [rng.choice([0.0, 0.40005245828215963, np.pi / 2, -np.pi]), 0.0, 0.0]
    pts = cloud_for(rng, np.argwhere(grid > 0), dims, pos, zyx, n, res)
    kw = dict(n_yaw=int(rng.choice([1, 3, 5, 7, 16])), yaw_k0=int(rng.integers(-8, 3)), yaw_kstep=int(rng.integers(0, 5)),
              n_t=int(rng.choice([1, 2, 5, 6, 8])), length=float(rng.choice([0.25, 0.05, 0.4, 0.3])))
    want, wbest = O.scan_vote(grid, res, pts, pos, zyx, **kw)
    h = shared(dims, n, res)
    try:
        h.write_map(0, grid.reshape(-1))
        got, gbest = h.scan_vote(pts, pos, zyx, **kw)
        note("votes"); note("votes with a winner", wbest >= 0); note("vote counts", int(want.sum()))
        if not (np.array_equal(got, want) and gbest == wbest):
            why.append(f"vote dims={dims} n={n} res={res} {kw}: {int((got != want).sum())} sums differ, best {gbest}/{wbest}")
        if n:                                                  # the same points as rows pick[] of a larger cloud on the device
            big = rng.normal(0, 2, (n + int(rng.integers(0, 50)), 3))
            pick = rng.permutation(len(big))[:n].astype(np.int32)
            big[pick] = pts
            dev = torch.tensor(big.reshape(-1), device="cuda")
            got2, gbest2 = h.scan_vote_device(dev.data_ptr(), len(big), pick, pos, zyx, **kw)
            note("votes on a device cloud")
            if not (np.array_equal(got2, want) and gbest2 == wbest):
                why.append(f"vote_device dims={dims} n={n} {kw}: {int((got2 != want).sum())} sums differ")
        if n <= 4096 and kw["n_yaw"] <= 16 and kw["n_t"] <= 8:
            g = torch.tensor(grid.reshape(-1), device="cuda"); p = torch.tensor(np.ascontiguousarray(pts).reshape(-1), device="cuda")
            s = torch.zeros(kw["n_yaw"] * kw["n_t"] ** 3, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            capi.scan_vote_static(0, 0, g.data_ptr(), dims, res, p.data_ptr() if n else 0, n, pos, zyx, s.data_ptr(), **kw)
            torch.cuda.synchronize()
            if not np.array_equal(s.cpu().numpy(), want):
                why.append(f"vote_static dims={dims} n={n} {kw}: {int((s.cpu().numpy() != want).sum())} sums differ")
    finally:
        h.close()


def one_ring(rng, why):
    dims = tuple(int(v) for v in rng.integers(40, 130, 3))
    cap = int(rng.choice([50, 700, 3000]))
    n_slots = int(rng.choice([1, 2, 5, 8]))
    move = float(rng.choice([0.15, 0.05, 0.4])); rxy = float(rng.choice([2.0, 0.5])); rz = float(rng.choice([1.0, 0.3]))
    life_new = int(rng.choice([6, 2, 12]))
    room = np.zeros(dims, np.uint8); room[rng.random(dims) < 0.03] = 1
    occ = np.argwhere(room > 0)
    ring = O.RingState(dims, 0.05, n_slots=n_slots, cap=cap)
    ring.r.move_threshold = move; ring.r.recentre_xy = rxy; ring.r.recentre_z = rz; ring.r.life_new = life_new
    grid = np.zeros(dims, np.uint8)
    pos = np.array(dims) * 0.05 / 2 + rng.normal(0, 0.1, 3)
    yaw = float(rng.uniform(-3, 3))
    vel = rng.normal(0, 0.08, 3)
    h = shared(dims, cap + 40)
    try:
        h.fill_maps(0)
        h.ring_init(n_slots=n_slots, slot_capacity=cap, move_threshold=move, recentre_xy=rxy, recentre_z=rz, life_new=life_new)
        for t in range(14):
            pos = snap(rng, pos + vel + rng.normal(0, 0.01, 3), 0.3); yaw += 0.04
            zyx = np.array([yaw, rng.normal(0, 0.01), rng.normal(0, 0.01)])
            n = int(rng.integers(0, cap + 30)) if rng.random() < 0.85 else 0
            cloud = cloud_for(rng, occ, dims, pos, zyx, n)
            ring.paint(grid); h.ring_paint()
            if not np.array_equal(h.read_map(0).reshape(dims), grid):
                why.append(f"ring dims={dims} cap={cap} slots={n_slots}: grid after paint, frame {t}"); return
            try:
                if n and rng.random() < 0.5:                       # the cloud read where it is on the device
                    dev = torch.tensor(np.ascontiguousarray(cloud).reshape(-1), device="cuda")
                    gf, gpos = h.ring_update_device(dev.data_ptr(), n, pos, zyx)
                    note("ring frames from a device cloud")
                else:
                    gf, gpos = h.ring_update(cloud, pos, zyx)
            except capi.PfError as e:
                if n > cap:                                        # the library refuses what the reference would overrun
                    continue
                why.append(f"ring: {e}"); return
            if n > cap:
                why.append(f"ring: {n} points into slots of {cap} accepted"); return
            of, opos = ring.update(grid, cloud, pos, zyx)
            counts, lives, pu, eu, idx = h.ring_state(with_idx=True)
            ok = (gf == of and np.array_equal(gpos, opos) and np.array_equal(counts, ring.count) and np.array_equal(lives, ring.life)
                  and np.array_equal(pu, np.array(ring.r.pos_update)) and np.array_equal(eu, np.array(ring.r.euler_update))
                  and np.array_equal(idx.reshape(-1), ring.idx) and np.array_equal(h.read_map(0).reshape(dims), grid))
            if not ok:
                why.append(f"ring dims={dims} cap={cap} slots={n_slots} move={move}: frame {t} flags {gf}/{of} counts {counts.tolist()}/{ring.count.tolist()}"); return
            note("ring frames"); note("ring replacements", of & 1); note("ring recentrings", (of & 2) // 2); note("ring cells painted", int(grid.sum()))
            if of & 2:
                pos = opos
    finally:
        h.close()


def one_depth(rng, why):
    pool = int(rng.choice([1, 2, 3, 4, 5, 8, 16]))
    width = int(rng.integers(pool, 700)); height = int(rng.integers(pool, 500))
    img = rng.integers(0, 65536, (height, width)).astype(np.uint16)
    kind = rng.random()
    if kind < 0.5:
        img = rng.integers(300, 9000, (height, width)).astype(np.uint16)
    img[rng.random((height, width)) < rng.choice([0.0, 0.35, 0.95])] = 0
    raw = np.frombuffer(img.astype("<u2").tobytes(), np.uint8)
    focal = float(rng.choice([385.0 / pool, 96.25, 200.0, 0.7]))
    want = O.depth_to_cloud(raw, width, height, pool, focal)
    h = shared((16, 16, 16), 16)
    try:
        got = h.depth_to_cloud(raw, width, height, pool, focal)
        note("depth images"); note("cloud points", len(want))
        if not (got.shape == want.shape and np.array_equal(got, want)):
            why.append(f"depth {width}x{height} pool {pool} focal {focal}: {got.shape} vs {want.shape}")
        cap = (width // pool) * (height // pool)
        dev = torch.full((3 * cap + 3,), -7.0, dtype=torch.float64, device="cuda")
        n2 = h.depth_to_cloud_device(raw, width, height, dev.data_ptr(), cap, pool, focal)
        back = dev.cpu().numpy()
        if not (n2 == len(want) and np.array_equal(back[:3 * n2].reshape(-1, 3), want) and np.all(back[3 * cap:] == -7.0)):
            why.append(f"depth_device {width}x{height} pool {pool}: {n2} vs {len(want)}")
    finally:
        h.close()


def one_readout(rng, why):
    dims = tuple(int(v) for v in rng.integers(3, 150, 3))
    grid = rng.integers(0, 256, dims).astype(np.uint8)
    grid[rng.random(dims) < rng.choice([0.0, 0.7, 0.99])] = 50
    stride = int(rng.choice([1, 2, 3, 4, 7, 64, 1000])); thr = int(rng.choice([0, 50, 80, 254, 255]))
    want = O.read_occupied(grid, 0.05, thr, stride)
    h = shared(dims, 16)
    try:
        h.broadcast_map(grid)
        got, total = h.read_occupied(0, thr, stride)
        note("read-outs"); note("read-out points", len(want))
        capn = int(rng.integers(0, len(want) + 2))
        few, total2 = h.read_occupied(0, thr, stride, capacity=capn)
        if not (total == len(want) and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
                and total2 == total and np.array_equal(few.view(np.uint32), want[:capn].view(np.uint32))):
            why.append(f"readout dims={dims} stride={stride} thr={thr}: total {total}/{len(want)}")
    finally:
        h.close()


def one_local_update(rng, why):
    dims = tuple(int(v) for v in rng.integers(20, 150, 3))
    n = int(rng.choice([1, 5, 64, 500, 3000])); r = int(rng.choice([0, 1, 100, 2400, 20000]))
    res = pick_res(rng)
    pos = snap(rng, np.array(dims) * res * rng.uniform(0.2, 0.8, 3), 0.3)
    zyx = np.array([rng.uniform(-3.2, 3.2), rng.normal(0, 0.03), rng.normal(0, 0.03)])
    d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    rngs = rng.integers(1, 140, n) * (res / 2)                     # on half-voxel steps (to 3.5 m at 0.05 m)
    pts = d * rngs[:, None]
    axis = rng.random(n) < 0.3
    pts[axis] = (np.eye(3)[rng.integers(0, 3, n)] * rngs[:, None])[axis]
    ridx = rng.integers(0, n, r).astype(np.int32)
    prm = (int(rng.choice([10, 1, 127])), int(rng.choice([50, 127, 0])), int(rng.choice([-3, -1, -127])), int(rng.choice([-50, -128, 0])))
    mr = float(rng.choice([3.0, 1.0, 100.0]))
    grid = rng.integers(-128, 128, dims).astype(np.int8) if rng.random() < 0.5 else np.zeros(dims, np.int8)
    cfg = O.UpdateCfg(dims[0], dims[1], dims[2], res, 1, prm[0], prm[1], prm[2], prm[3], mr)
    want = grid.copy(); O.update_map(cfg, want, pos, zyx, pts, ridx)
    h = capi.Handle(capi.default_config(n_particles=1, grid_x=8, grid_y=8, grid_z=8, local_x=dims[0], local_y=dims[1], local_z=dims[2],
                                        max_beams=16, max_points=max(n, r, 1), resolution=res))
    try:
        h.write_local_map(grid)
        if rng.random() < 0.5:
            dev = torch.tensor(np.ascontiguousarray(pts).reshape(-1), device="cuda")
            h.set_scan_device(dev.data_ptr(), n, ridx); note("local updates from a device cloud")
        else:
            h.set_scan(pts, ridx)
        h.update_map(capi.PF_GRID_LOCAL, capi.UpdateParams(*prm, mr), pos, zyx)
        got = h.read_local_map().reshape(dims)
        note("local updates"); note("local cells changed", int((want != grid).sum()))
        if not np.array_equal(got, want):
            why.append(f"local update dims={dims} n={n} r={r} prm={prm} range={mr} res={res}: {int((got != want).sum())} cells")
    finally:
        h.close()


KINDS = [one_vote, one_ring, one_depth, one_readout, one_local_update]
for case in range(cases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 104729 + case)
    why = []
    fn = KINDS[case % len(KINDS)]
    try:
        fn(rng, why)
    except Exception as e:                                         # a refusal the oracle does not share is a finding too
        why.append(f"{fn.__name__}: {type(e).__name__}: {e}")
    if why:
        bad += 1
        print(f"case {case}: MISMATCH " + "; ".join(why), flush=True)
print("exercised:", ", ".join(f"{k} {v}" for k, v in seen.items()), flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
