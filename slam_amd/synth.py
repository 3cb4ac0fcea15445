"""Synthetic inputs for tests and bench: a box room with obstacles rasterised at
the grid resolution, scans ray-traced from a pose, and the beam list a host driver would hand
to pf_set_beams. Deterministic for a given seed. Host-side numpy only; nothing here is on the
measured path."""
import numpy as np


def make_room(dims, seed=0, margin=None):
    """uint8 occupancy grid [gx, gy, gz]: 50 unknown outside, 0 free inside, 100 walls, floor,
    ceiling and a few box obstacles."""
    gx, gy, gz = dims
    rng = np.random.default_rng(seed)
    m = np.full(dims, 50, np.uint8)
    mx = margin if margin is not None else max(2, gx // 12)
    my = margin if margin is not None else max(2, gy // 12)
    mz = max(1, gz // 12)
    x0, x1, y0, y1, z0, z1 = mx, gx - 1 - mx, my, gy - 1 - my, mz, gz - 1 - mz
    m[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0
    m[x0, y0:y1 + 1, z0:z1 + 1] = 100; m[x1, y0:y1 + 1, z0:z1 + 1] = 100
    m[x0:x1 + 1, y0, z0:z1 + 1] = 100; m[x0:x1 + 1, y1, z0:z1 + 1] = 100
    m[x0:x1 + 1, y0:y1 + 1, z0] = 100; m[x0:x1 + 1, y0:y1 + 1, z1] = 100
    for _ in range(6):
        sx, sy = rng.integers(gx // 16 + 1, gx // 8 + 2, 2)
        sz = int(rng.integers(gz // 6 + 1, gz // 2 + 2))
        ox = int(rng.integers(x0 + 1, max(x0 + 2, x1 - sx)))
        oy = int(rng.integers(y0 + 1, max(y0 + 2, y1 - sy)))
        # keep the centre of the room free for the sensor
        if abs(ox + sx / 2 - gx / 2) < gx / 8 + sx / 2 and abs(oy + sy / 2 - gy / 2) < gy / 8 + sy / 2:
            continue
        m[ox:ox + sx, oy:oy + sy, z0:min(z1, z0 + sz)] = 100
    return m


def trace_scan(grid, pose_vox, yaw, n_rays, max_range_vox, seed=1):
    """Ray-trace n_rays directions from pose_vox (voxel units, grid frame) to the first cell with
    occupancy 100. Returns endpoints in the SENSOR frame in voxel units, float64 [k, 3] (rays
    that hit nothing within range are dropped)."""
    rng = np.random.default_rng(seed)
    az = rng.uniform(-np.pi, np.pi, n_rays)
    el = rng.uniform(-0.6, 0.6, n_rays)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    c, s = np.cos(yaw), np.sin(yaw)
    dw = np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1], d[:, 2]], 1)
    gx, gy, gz = grid.shape
    hit_t = np.full(n_rays, np.nan)
    alive = np.ones(n_rays, bool)
    p0 = np.asarray(pose_vox, np.float64)
    flat = grid.reshape(-1)
    for t in np.arange(1.0, max_range_vox, 0.5):
        idx = np.flatnonzero(alive)
        if idx.size == 0:
            break
        q = np.rint(p0 + dw[idx] * t).astype(np.int64)
        inside = (q[:, 0] >= 0) & (q[:, 0] < gx) & (q[:, 1] >= 0) & (q[:, 1] < gy) & \
                 (q[:, 2] >= 0) & (q[:, 2] < gz)
        occ = np.zeros(idx.size, bool)
        f = (q[inside, 0] * gy + q[inside, 1]) * gz + q[inside, 2]
        occ[inside] = flat[f] == 100
        hit_t[idx[occ]] = t
        alive[idx[occ]] = False
        alive[idx[~inside]] = False
    ok = ~np.isnan(hit_t)
    return d[ok] * hit_t[ok, None]


def beams_from_endpoints(end_vox, n_beams, value=50, seed=2):
    """Distinct local voxels of the endpoints (sensor frame, voxel units), at most n_beams of
    them, in ascending (x, y, z) order — the order of S3P:868 — with the given int8 value."""
    v = np.unique(np.rint(end_vox).astype(np.int32), axis=0)       # sorted lexicographically
    if len(v) > n_beams:
        rng = np.random.default_rng(seed)
        keep = np.sort(rng.choice(len(v), n_beams, replace=False))
        v = v[keep]
    bv = np.full(len(v), value, np.int8)
    return v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy(), bv


def scene(dims, n_particles, n_beams, resolution=0.05, seed=0, sigma_pos=0.03, sigma_yaw=0.02,
          max_range_vox=None):
    """A whole synthetic workload: grid, true pose, particle cloud around it, beam list, the scan
    in metres (sensor frame) and ray indices for the map update."""
    gx, gy, gz = dims
    grid = make_room(dims, seed)
    true_vox = np.array([gx / 2 + 0.3, gy / 2 - 0.2, gz / 2 + 0.1])
    yaw = 0.35
    rng = np.random.default_rng(seed + 100)
    mr = max_range_vox or int(1.2 * max(gx, gy))
    n_rays = 3 * n_beams
    ends = trace_scan(grid, true_vox, yaw, n_rays, mr, seed + 1)
    while True:
        b = beams_from_endpoints(ends, n_beams, 50, seed + 2)
        if len(b[0]) >= n_beams or n_rays > 64 * n_beams:
            break
        n_rays *= 2
        ends = trace_scan(grid, true_vox, yaw, n_rays, mr, seed + 1)
    P = n_particles
    x = true_vox[0] * resolution + rng.normal(0, sigma_pos, P)
    y = true_vox[1] * resolution + rng.normal(0, sigma_pos, P)
    z = true_vox[2] * resolution + rng.normal(0, sigma_pos * 0.5, P)
    yw = yaw + rng.normal(0, sigma_yaw, P)
    pts = ends[:min(len(ends), n_beams)] * resolution
    return dict(grid=grid, true_pose=np.array([*(true_vox * resolution), yaw]), x=x, y=y, z=z,
                yaw=yw, beams=b, points=pts, ray_idx=np.arange(len(pts), dtype=np.int32))
