#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — fixtures for the per-frame half of the path (depth front-end, scan vote,
keyframe ring, local-map update) and for the whole callback + particle_map loop, produced by
RUNNING THE REFERENCE'S OWN class Slam (oracle/_ref/libs3p_ref.so, built by oracle/build_ref.sh
from the text of /root/reference/Slam_3d_particle.cpp: members S3P:115-146, 150-151, table loops
S3P:157-162, member functions S3P:258-729, free functions S3P:12-110, 731-962).

Runs only in the build container. The fixtures hold data only: the synthetic inputs made here
(depth images, clouds, IMU angles, seeds), the libstdc++ draws the reference consumed (as plain
index arrays / CRCs) and what the reference's functions left in its members and globals.

    python oracle/make_golden_frontend.py      # writes tests/golden/{depth,vote,local_map,ring,frames,helpers_p}.npz

Stale-row screening: S3P:317 and S3P:666 draw indices in [0, cloudpoints_index] INCLUSIVE; index
== cloudpoints_index reads a row that holds whatever an earlier frame left there. Seeds whose
stream contains that index are skipped (the seed actually used is stored in the fixture).
"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(HERE, "_ref", "libs3p_ref.so")

LX, LY, LZ = 200, 200, 80
NL = LX * LY * LZ
RES_H, RES_V = 160, 120
FOCAL = 385.0 / 4.0
SLOTS, CAP = 5, RES_H * RES_V


def load_ref():
    r = C.CDLL(REF)
    r.ref_particle_map_ptr.restype = C.POINTER(C.c_uint8)
    r.ref_map_local_ptr.restype = C.POINTER(C.c_int8)
    r.ref_map_localization_ptr.restype = C.POINTER(C.c_uint8)
    r.ref_ring_idx.restype = C.POINTER(C.c_int)
    r.ref_ring_count.restype = C.POINTER(C.c_int)
    r.ref_ring_life.restype = C.POINTER(C.c_int)
    r.ref_set_particle.argtypes = [C.c_int] + [C.c_double] * 5
    r.ref_set_local_pose.argtypes = [C.c_double] * 4
    r.ref_set_global_pose.argtypes = [C.c_double] * 4
    r.ref_set_policy.argtypes = [C.c_int] + [C.c_double] * 4 + [C.c_int]
    r.ref_set_gyr.argtypes = [C.c_double] * 3
    r.ref_set_scan_pose.argtypes = [C.c_double] * 4
    r.ref_set_diffs.argtypes = [C.c_double] * 4
    r.ref_set_rot_yx.argtypes = [C.c_double] * 2
    r.ref_set_feedback.argtypes = [C.c_double] * 4
    assert r.ref_res_h() == RES_H and r.ref_res_v() == RES_V
    return r


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def dvec(r, fn, n):
    buf = (C.c_double * n)()
    getattr(r, fn)(buf)
    return np.array(buf[:])


def loc_grid(r):
    return np.ctypeslib.as_array(r.ref_map_localization_ptr(), shape=(NL,))


def local_grid(r):
    return np.ctypeslib.as_array(r.ref_map_local_ptr(), shape=(NL,))


def ring_arrays(r):
    return (np.ctypeslib.as_array(r.ref_ring_idx(), shape=(SLOTS * CAP,)),
            np.ctypeslib.as_array(r.ref_ring_count(), shape=(SLOTS,)),
            np.ctypeslib.as_array(r.ref_ring_life(), shape=(SLOTS,)))


def index_stream(r, seed, n, hi):
    out = np.empty(n, np.int32)
    r.ref_index_stream(C.c_uint(seed), n, hi, vp(out))
    return out


def screened_seed(r, seed, n, hi):
    """First seed >= `seed` whose n draws of uniform_int(0, hi) avoid the stale row hi."""
    for _ in range(5000):
        idx = index_stream(r, seed, n, hi)
        if not np.any(idx == hi):
            return seed, idx
        seed += 1
    raise RuntimeError(f"no seed avoids the stale row: n={n} hi={hi}")


# ---------------------------------------------------------------------------------------------
# synthetic scene: a box room with box obstacles, seen by the reference's pinhole camera
# ---------------------------------------------------------------------------------------------
ROOM = (2.23, 7.91, 2.62, 8.37, 0.71, 3.43)               # x0 x1 y0 y1 z0 z1 in the localisation frame
BOXES = [(6.1, 6.9, 6.4, 7.5, 0.71, 2.2), (3.0, 3.8, 7.2, 8.37, 0.71, 1.6), (2.23, 2.9, 4.1, 5.3, 1.2, 2.6),
         (5.2, 5.6, 7.6, 7.9, 0.71, 3.43)]


def rot_zyx_np(z, y, x):
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    return np.array([[cy * cz, cz * sx * sy - cx * sz, sx * sz + cx * cz * sy],
                     [cy * sz, cx * cz + sx * sy * sz, cx * sy * sz - cz * sx],
                     [-sy, cy * sx, cx * cy]])


def pooled_depth(pose, zyx, frame_no, max_mm=6000):
    """uint16 [120][160] pooled depths (mm along the camera's y axis) of the scene from `pose`."""
    jj = (np.arange(RES_H) + 1 - RES_H // 2) / FOCAL
    ii = -1 * (np.arange(RES_V) + 1 - RES_V // 2) / FOCAL
    d_cam = np.stack([np.broadcast_to(jj[None, :], (RES_V, RES_H)), np.ones((RES_V, RES_H)),
                      np.broadcast_to(ii[:, None], (RES_V, RES_H))], -1).reshape(-1, 3)
    d = d_cam @ rot_zyx_np(*zyx).T
    o = np.asarray(pose, np.float64)
    t = np.full(len(d), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # room walls, from inside
        for ax in range(3):
            lo, hi = ROOM[2 * ax], ROOM[2 * ax + 1]
            tt = np.where(d[:, ax] > 0, (hi - o[ax]) / d[:, ax], np.where(d[:, ax] < 0, (lo - o[ax]) / d[:, ax], np.inf))
            t = np.minimum(t, np.where(tt > 0, tt, np.inf))
        # obstacles, from outside (slab test)
        for b in BOXES:
            tn = np.full(len(d), -np.inf); tf = np.full(len(d), np.inf)
            for ax in range(3):
                t1 = (b[2 * ax] - o[ax]) / d[:, ax]; t2 = (b[2 * ax + 1] - o[ax]) / d[:, ax]
                tn = np.maximum(tn, np.minimum(t1, t2)); tf = np.minimum(tf, np.maximum(t1, t2))
            hit = (tn <= tf) & (tn > 0)
            t = np.minimum(t, np.where(hit, tn, np.inf))
    mm = np.rint(t * 1000.0)
    k = np.arange(len(d), dtype=np.int64)
    h = (k * 2654435761 + frame_no * 40503) % 4294967296                 # integer hash: no RNG state
    mm = mm + ((h >> 7) % 5) - 2                                         # +-2 mm sensor noise
    mm = np.where((h >> 13) % 23 == 0, 0, mm)                            # dropouts
    mm = np.where((mm > max_mm) | ~np.isfinite(mm) | (mm < 150), 0, mm)
    return mm.reshape(RES_V, RES_H).astype(np.uint16)


def full_image(pooled):
    return np.repeat(np.repeat(pooled, 4, 0), 4, 1)


# ---------------------------------------------------------------------------------------------
def gen_helpers_p(r, rng):
    js = np.concatenate([rng.integers(0, 300 * 300 * 80, 400), [0, 79, 80, 23999, 24000, 7199999, 7199996]]).astype(np.int32)
    rev = np.empty((len(js), 3), np.int32); tmp = (C.c_int * 3)()
    for k, j in enumerate(js):
        r.ref_reverse_index_conv_p(tmp, int(j)); rev[k] = tmp[:]
    np.savez_compressed(os.path.join(OUT, "helpers_p.npz"), js_p=js, rev_p=rev)
    print("helpers_p:", len(js), "indices")


def gen_depth(r, rng):
    """resize_depth_to_array + img_to_cloud (S3P:562-612) on one full 640x480 image with uneven
    4x4 blocks: zeros inside blocks, all-zero blocks, saturated pixels, sums that do not divide."""
    r.ref_slam_reset()
    pooled = pooled_depth((5.0, 5.0, 2.0), (0.3, 0.02, -0.04), 7)
    img = full_image(pooled).astype(np.int64)
    img += rng.integers(-3, 4, img.shape)                                 # mean no longer trivial
    img[rng.random(img.shape) < 0.08] = 0                                 # zeros inside blocks
    img[np.repeat(np.repeat(pooled == 0, 4, 0), 4, 1)] = 0                # empty blocks stay empty
    img[40:44, 100:104] = 65535; img[44:48, 100:104] = 0; img[45, 101] = 65535
    img[200:204, 300:304] = 0; img[201, 302] = 1
    img = np.clip(img, 0, 65535).astype("<u2")
    raw = np.frombuffer(img.tobytes(), np.uint8)
    n = r.ref_depth_to_cloud(vp(raw), len(raw))
    cloud = np.empty((RES_H * RES_V, 3)); r.ref_get_cloud(vp(cloud), len(cloud))
    dx = np.empty((RES_V, RES_H), np.int32); dy = dx.copy(); dz = dx.copy()
    r.ref_get_depth_arrays(vp(dx), vp(dy), vp(dz))
    np.savez_compressed(os.path.join(OUT, "depth.npz"), image=img, n=np.int32(n), cloud_mm=mm16(cloud[:n]),
                        depth_x=dx, depth_y=dy, depth_z=dz)
    print(f"depth: {n} points of {RES_H * RES_V}")


def synthetic_cloud(rng, n, spread=2.5):
    p = np.empty((n, 3))
    p[:, 1] = rng.uniform(0.4, 3.5, n)
    p[:, 0] = p[:, 1] * rng.uniform(-0.8, 0.8, n)
    p[:, 2] = p[:, 1] * rng.uniform(-0.6, 0.6, n)
    return np.round(p * 1000) / 1000.0      # millimetre lattice like img_to_cloud's output


def mm16(cloud):
    """Clouds are stored as integer (int16 when they fit) millimetres: cloud == mm / 1000.0 exactly (S3P:603-605)."""
    mm = np.rint(cloud * 1000.0)
    assert np.array_equal(mm / 1000.0, cloud)
    return mm.astype(np.int16 if np.abs(mm).max() < 32768 else np.int32)


PRE_TABLE = np.array([-50, -49, -48, -3, 0, 0, 0, 10, 41, 45, 50], np.int8)


def pre_map(k):
    """Input local map of local_map case k: an integer hash of the cell index (no RNG state), so
    the fixture need not store 3.2 MB per case. tests/test_oracle_golden.py holds the same lines."""
    j = np.arange(NL, dtype=np.int64)
    return PRE_TABLE[((j * 2654435761 + k * 97) >> 11) % len(PRE_TABLE)]


def gen_vote(r, rng):
    """scan_localization() alone (S3P:300-410): grid, cloud, scan pose and IMU angles set by the
    harness; the outputs are the members the function leaves behind."""
    cases = {}
    k = 0
    for kind in ("match", "match", "match_tilt", "match_tilt", "edge", "empty_grid", "wrap_pos", "wrap_neg",
                 "sparse", "after_previous"):
        if kind != "after_previous":
            r.ref_slam_reset()
        n = int(rng.integers(700, 5000))
        cloud = synthetic_cloud(rng, n)
        rot_y, rot_x = (0.0, 0.0) if "tilt" not in kind else rng.uniform(-0.12, 0.12, 2)
        if kind == "edge":
            pos = np.array([9.3, 0.6, 3.7]); yaw = 2.9
        else:
            pos = np.array([5.0, 5.0, 2.0]) + rng.uniform(-1.5, 1.5, 3) * [1, 1, 0.3]; yaw = rng.uniform(-3.0, 3.0)
        if kind == "after_previous":
            st = dvec(r, "ref_get_scan_state", 18); pos = st[0:3].copy(); yaw = st[3]
        # the grid: the cloud seen from a pose a little off the scan pose, plus clutter
        off = rng.integers(-2, 3, 3) * 0.05 + rng.uniform(-0.02, 0.02, 3)
        dyaw = np.deg2rad(rng.choice([-6, -3, 0, 3, 6])) + rng.uniform(-0.01, 0.01)
        w = cloud @ rot_zyx_np(yaw + dyaw, rot_y, rot_x).T + pos + off
        idx = np.rint(w * 20).astype(np.int64)
        flat = idx[:, 0] * LY * LZ + idx[:, 1] * LZ + idx[:, 2]
        flat = flat[(flat >= 0) & (flat < NL)]
        if kind == "sparse":
            flat = flat[::40]
        if kind == "empty_grid":
            flat = flat[:0]
        clutter = rng.integers(0, NL, 3000)
        g = loc_grid(r)
        if kind != "after_previous":
            g[:] = 0
            g[flat] = 1; g[clutter] = 1
        grid_idx = np.flatnonzero(g).astype(np.int32)
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_scan_pose(*[float(v) for v in pos], float(yaw))
        if kind == "wrap_pos":
            gyr = (3.1, rot_y, rot_x); gyr_old_first = -3.1
        elif kind == "wrap_neg":
            gyr = (-3.1, rot_y, rot_x); gyr_old_first = 3.1
        else:
            gyr = (yaw + rng.uniform(-0.05, 0.05), rot_y, rot_x); gyr_old_first = None
        if gyr_old_first is not None:
            # one call with the old angle so that euler_zyx_diff_gyr_old holds it (S3P:403)
            r.ref_set_gyr(float(gyr_old_first), float(rot_y), float(rot_x))
            s0, _ = screened_seed(r, 90000 + 17 * k, 500, n); r.ref_set_seed(s0); r.ref_scan_localization()
            r.ref_set_scan_pose(*[float(v) for v in pos], float(yaw))
        before = dvec(r, "ref_get_scan_state", 18)
        r.ref_set_gyr(*[float(v) for v in gyr])
        seed, pick = screened_seed(r, 70000 + 31 * k, 500, n)
        r.ref_set_seed(seed)
        r.ref_scan_localization()
        after = dvec(r, "ref_get_scan_state", 18)
        test = np.empty((500, 3)); r.ref_get_cloud_test(vp(test))
        assert np.array_equal(test, cloud[pick]), "index stream does not reproduce cloud_test"
        name = f"c{k}_{kind}"
        cases[name + "_grid_idx"] = grid_idx; cases[name + "_cloud_mm"] = mm16(cloud)
        cases[name + "_gyr"] = np.array(gyr, np.float64); cases[name + "_seed"] = np.int64(seed)
        cases[name + "_pick"] = pick; cases[name + "_before"] = before; cases[name + "_after"] = after
        print(f"vote {name}: n={n} loc={after[4:8]} last sum_={after[17]:.0f}")
        k += 1
    np.savez_compressed(os.path.join(OUT, "vote.npz"), **cases)


def gen_local_map(r, rng):
    """build_local_map() alone (S3P:621-710) on prepared inputs; output: every cell of map_local
    that differs from the input map, the updated local pose and the dead-reckoned global pose."""
    cases = {}
    for k, kind in enumerate(("fresh", "saturated", "tilt", "far_and_edge", "small_cloud")):
        r.ref_slam_reset()
        n = {"small_cloud": 700}.get(kind, int(rng.integers(3000, 12000)))
        cloud = synthetic_cloud(rng, n)
        if kind == "far_and_edge":
            cloud[: n // 3] *= 2.2                     # beyond 3 m: no hit, but still ray targets (S3P:677-679 is dead code)
            cloud = np.round(cloud * 1000) / 1000.0
        pre = np.zeros(NL, np.int8)
        if kind in ("saturated", "tilt", "far_and_edge"):
            pre[:] = pre_map(k)
        local_pose = np.array([5.0, 5.0, 1.5, 0.0]) + (rng.uniform(-0.3, 0.3, 4) if kind != "far_and_edge" else
                                                        np.array([4.2, -4.1, 1.9, 2.5]))
        diffs = rng.uniform(-0.06, 0.06, 4)
        rot_y, rot_x = rng.uniform(-0.1, 0.1, 2) if kind in ("tilt", "far_and_edge") else (0.0, 0.0)
        gpose = np.array([7.5, 7.5, 1.5, 0.0]) + rng.uniform(-1, 1, 4)
        if kind == "tilt":
            gpose[3] = 3.13                            # the wrap of S3P:717-722
            diffs[3] = 0.05
        fb = rng.uniform(-0.05, 0.05, 4)
        local_grid(r)[:] = pre
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_local_pose(*[float(v) for v in local_pose])
        r.ref_set_global_pose(*[float(v) for v in gpose])
        r.ref_set_feedback(*[float(v) for v in fb])
        r.ref_set_diffs(*[float(v) for v in diffs])
        r.ref_set_rot_yx(float(rot_y), float(rot_x))
        seed, rays = screened_seed(r, 50000 + 13 * k, 2400, n)
        r.ref_set_seed(seed)
        r.ref_build_local_map()
        post = local_grid(r).copy()
        touched = np.flatnonzero(post != pre).astype(np.int32)
        lp = dvec(r, "ref_get_local_pose", 4); gp = dvec(r, "ref_get_global_pose", 4)
        fb_after = dvec(r, "ref_get_feedback", 4)
        name = f"m{k}_{kind}"
        cases[name + "_cloud_mm"] = mm16(cloud)
        cases[name + "_pre_k"] = np.int64(k if pre.any() else -1)      # pre_map(k), or all zero
        cases[name + "_local_pose"] = local_pose; cases[name + "_diffs"] = diffs
        cases[name + "_rot_yx"] = np.array([rot_y, rot_x]); cases[name + "_gpose"] = gpose; cases[name + "_fb"] = fb
        cases[name + "_seed"] = np.int64(seed); cases[name + "_rays"] = rays
        cases[name + "_touched"] = touched; cases[name + "_touched_val"] = post[touched]
        cases[name + "_map_crc"] = crc(post)
        cases[name + "_local_pose_after"] = lp; cases[name + "_gpose_after"] = gp; cases[name + "_fb_after"] = fb_after
        print(f"local_map {name}: n={n} touched={len(touched)}")
    np.savez_compressed(os.path.join(OUT, "local_map.npz"), **cases)


def gen_local_map_ties(r):
    """build_local_map() where its roundings sit on ties (own RNG: the other fixtures do not move). Scan
    points at odd multiples of 25 mm from the sensor along directions with integer components (3-4-5,
    7-24-25, 12-16-15, 9-12-20 and the axes), so that round(norm * 20) (S3P:676) is on a tie in exact
    arithmetic, and at exactly 3 m for the range test (S3P:637); the local yaws are angles at which glibc's
    sincos() -- which g++ -O2 makes of the reference's sin() and cos() calls -- differs from the separate
    calls in the last place. What decides each tie is then the last place of the rotated point."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.cos.restype = libm.sin.restype = C.c_double
    libm.cos.argtypes = libm.sin.argtypes = [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(7700)
    yaws = [0.40005245828215963]
    while len(yaws) < 4:
        a = float(rng.uniform(-3.1, 3.1))
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(sv), C.byref(cv))
        if cv.value != libm.cos(a) and sv.value != libm.sin(a):
            yaws.append(a)
    dirs = []
    for t in ((1, 0, 0), (3, 4, 0), (7, 24, 0), (12, 16, 15), (9, 12, 20)):
        q = np.array(t) * (25 // int(round(np.sqrt(sum(v * v for v in t)))))      # |q| = 25 (mm)
        for perm in ((0, 1, 2), (1, 0, 2), (2, 1, 0), (0, 2, 1), (1, 2, 0), (2, 0, 1)):
            for sg in ((1, 1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, -1), (-1, -1, 1), (-1, 1, -1), (1, -1, -1), (-1, -1, -1)):
                dirs.append(q[list(perm)] * np.array(sg))
    dirs = np.unique(np.array(dirs), axis=0)
    cases = {}
    for k, yaw in enumerate(yaws):
        r.ref_slam_reset()
        n = 4000
        m = rng.integers(2, 70, n) * 2 + 1                       # odd: the step count is on a tie
        m[::9] = 120                                             # exactly 3 m: the range test is
        m[1::9] = rng.integers(2, 70, len(m[1::9])) * 2          # even: whole steps, the plain case
        cloud = (dirs[rng.integers(0, len(dirs), n)] * m[:, None]) / 1000.0
        pre = np.zeros(NL, np.int8)
        if k % 2:
            pre[:] = pre_map(k)
        local_pose = np.array([5.0 + 0.025 * int(rng.integers(-8, 9)), 5.0 + 0.025 * int(rng.integers(-8, 9)), 1.5, yaw])
        if k >= 2:
            local_pose[:3] += rng.uniform(-0.2, 0.2, 3)
        diffs = np.zeros(4)
        rot_y, rot_x = (0.0, 0.0) if k < 3 else (0.013, -0.021)
        gpose = np.array([7.5, 7.5, 1.5, yaw])
        fb = np.zeros(4)
        local_grid(r)[:] = pre
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_local_pose(*[float(v) for v in local_pose])
        r.ref_set_global_pose(*[float(v) for v in gpose])
        r.ref_set_feedback(*[float(v) for v in fb])
        r.ref_set_diffs(*[float(v) for v in diffs])
        r.ref_set_rot_yx(float(rot_y), float(rot_x))
        seed, rays = screened_seed(r, 52000 + 13 * k, 2400, n)
        r.ref_set_seed(seed)
        r.ref_build_local_map()
        post = local_grid(r).copy()
        touched = np.flatnonzero(post != pre).astype(np.int32)
        lp = dvec(r, "ref_get_local_pose", 4); gp = dvec(r, "ref_get_global_pose", 4)
        fb_after = dvec(r, "ref_get_feedback", 4)
        name = f"t{k}_ties"
        cases[name + "_cloud_mm"] = mm16(cloud)
        cases[name + "_pre_k"] = np.int64(k if pre.any() else -1)
        cases[name + "_local_pose"] = local_pose; cases[name + "_diffs"] = diffs
        cases[name + "_rot_yx"] = np.array([rot_y, rot_x]); cases[name + "_gpose"] = gpose; cases[name + "_fb"] = fb
        cases[name + "_seed"] = np.int64(seed); cases[name + "_rays"] = rays
        cases[name + "_touched"] = touched; cases[name + "_touched_val"] = post[touched]
        cases[name + "_map_crc"] = crc(post)
        cases[name + "_local_pose_after"] = lp; cases[name + "_gpose_after"] = gp; cases[name + "_fb_after"] = fb_after
        print(f"local_map_ties {name}: yaw={yaw!r} touched={len(touched)}")
    np.savez_compressed(os.path.join(OUT, "local_map_ties.npz"), **cases)


def gen_vote_trig(r):
    """scan_localization() at scan yaws where glibc's sincos() and sin()/cos() disagree in the last place (own
    RNG). The epilogue S3P:379-380 turns the voted offset by -yaw: g++ -O2 takes the pair of S3P:379 from one
    sincos(yaw) and calls sin(-yaw) again for S3P:380 (pf_host.cpp mirrors exactly that); with an offset of a
    few voxels the last place of pos_xyz_diff_scan tells the forms apart."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.cos.restype = libm.sin.restype = C.c_double
    libm.cos.argtypes = libm.sin.argtypes = [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(7800)
    want_kinds = ["sin", "cos", "both", "sin", "cos", "both"]
    yaws = []
    while len(yaws) < len(want_kinds):
        a = float(rng.uniform(-3.0, 3.0))
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(sv), C.byref(cv))
        ds, dc = sv.value != libm.sin(a) or -sv.value != libm.sin(-a), cv.value != libm.cos(a)
        kind = "both" if ds and dc else "sin" if ds else "cos" if dc else None
        if kind == want_kinds[len(yaws)]:
            yaws.append(a)
    cases = {}
    for k, yaw in enumerate(yaws):
        r.ref_slam_reset()
        n = int(rng.integers(1500, 4000))
        cloud = synthetic_cloud(rng, n)
        pos = np.array([5.0, 5.0, 2.0]) + rng.uniform(-1.0, 1.0, 3) * [1, 1, 0.3]
        off = np.array([2, -1, 1]) * 0.05 * rng.choice([-1, 1], 3) + rng.uniform(-0.01, 0.01, 3)    # a vote that moves the pose
        w = cloud @ rot_zyx_np(yaw, 0.0, 0.0).T + pos + off
        idx = np.rint(w * 20).astype(np.int64)
        flat = idx[:, 0] * LY * LZ + idx[:, 1] * LZ + idx[:, 2]
        flat = flat[(flat >= 0) & (flat < NL)]
        g = loc_grid(r)
        g[:] = 0; g[flat] = 1; g[rng.integers(0, NL, 2000)] = 1
        grid_idx = np.flatnonzero(g).astype(np.int32)
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_scan_pose(*[float(v) for v in pos], float(yaw))
        gyr = (yaw + rng.uniform(-0.05, 0.05), 0.0, 0.0)
        before = dvec(r, "ref_get_scan_state", 18)
        r.ref_set_gyr(*[float(v) for v in gyr])
        seed, pick = screened_seed(r, 71000 + 31 * k, 500, n)
        r.ref_set_seed(seed)
        r.ref_scan_localization()
        after = dvec(r, "ref_get_scan_state", 18)
        name = f"c{k}_trig"
        cases[name + "_grid_idx"] = grid_idx; cases[name + "_cloud_mm"] = mm16(cloud)
        cases[name + "_gyr"] = np.array(gyr, np.float64); cases[name + "_seed"] = np.int64(seed)
        cases[name + "_pick"] = pick; cases[name + "_before"] = before; cases[name + "_after"] = after
        print(f"vote_trig {name}: yaw={yaw!r} loc={after[4:8]} diff={after[8:11]}")
    np.savez_compressed(os.path.join(OUT, "vote_trig.npz"), **cases)


def gen_local_map_trig(r):
    """build_local_map() with the GLOBAL yaw at angles where glibc's sincos() and sin()/cos() disagree (own
    RNG): update_realtime_pos() (S3P:712-728) turns the scan-pose increment by it, from one sincos() in the
    g++ -O2 build; increments of a few metres let the last place reach the dead-reckoned pose."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.cos.restype = libm.sin.restype = C.c_double
    libm.cos.argtypes = libm.sin.argtypes = [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(8100)
    cases = {}
    k = 0
    while k < 8:
        a = float(rng.uniform(-3.0, 3.0))
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(sv), C.byref(cv))
        if sv.value == libm.sin(a) and cv.value == libm.cos(a):
            continue
        r.ref_slam_reset()
        n = 700
        cloud = synthetic_cloud(rng, n)
        pre = np.zeros(NL, np.int8)
        local_pose = np.array([5.0, 5.0, 1.5, 0.0]) + rng.uniform(-0.3, 0.3, 4)
        diffs = np.array([*rng.uniform(-2.5, 2.5, 2), rng.uniform(-0.05, 0.05), 0.0])
        gpose = np.array([0.0, 0.0, 1.5, a]) + np.array([*rng.uniform(-0.5, 0.5, 2), 0.0, 0.0])
        fb = np.zeros(4)
        local_grid(r)[:] = pre
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_local_pose(*[float(v) for v in local_pose])
        r.ref_set_global_pose(*[float(v) for v in gpose])
        r.ref_set_feedback(*[float(v) for v in fb])
        r.ref_set_diffs(*[float(v) for v in diffs])
        r.ref_set_rot_yx(0.0, 0.0)
        seed, rays = screened_seed(r, 53000 + 13 * k, 2400, n)
        r.ref_set_seed(seed)
        r.ref_build_local_map()
        post = local_grid(r).copy()
        touched = np.flatnonzero(post != pre).astype(np.int32)
        lp = dvec(r, "ref_get_local_pose", 4); gp = dvec(r, "ref_get_global_pose", 4)
        fb_after = dvec(r, "ref_get_feedback", 4)
        name = f"g{k}_trig"
        cases[name + "_cloud_mm"] = mm16(cloud)
        cases[name + "_pre_k"] = np.int64(-1)
        cases[name + "_local_pose"] = local_pose; cases[name + "_diffs"] = diffs
        cases[name + "_rot_yx"] = np.array([0.0, 0.0]); cases[name + "_gpose"] = gpose; cases[name + "_fb"] = fb
        cases[name + "_seed"] = np.int64(seed); cases[name + "_rays"] = rays
        cases[name + "_touched"] = touched; cases[name + "_touched_val"] = post[touched]
        cases[name + "_map_crc"] = crc(post)
        cases[name + "_local_pose_after"] = lp; cases[name + "_gpose_after"] = gp; cases[name + "_fb_after"] = fb_after
        # would separate calls have given another pose? (printed for the record)
        alt = gpose[0] + (libm.cos(a) * diffs[0] - libm.sin(a) * diffs[1] + 0.0)
        print(f"local_map_trig {name}: yaw={a!r} gpose_after={gp[:2]} separate-calls x differs: {alt != gp[0]}")
        k += 1
    np.savez_compressed(os.path.join(OUT, "local_map_trig.npz"), **cases)


def gen_ring(r, rng):
    """scan_localization_map_update() alone (S3P:419-529), poses set by the harness: keyframes whose
    points partly leave the grid (the non-compacted write of S3P:454-455), slot expiry, and
    recentrings whose flat-index offset throws entries out (S3P:502-516)."""
    r.ref_slam_reset()
    idx_arr, cnt_arr, life_arr = ring_arrays(r)
    poses = [(5.0, 5.0, 2.0, 0.0), (5.1, 5.0, 2.0, 0.02), (5.3, 5.1, 2.0, 0.1), (1.2, 5.1, 2.0, 1.6),
             (5.0, 5.0, 2.0, 0.3), (5.0, 5.05, 2.0, 0.3), (6.9, 6.9, 2.9, -0.5), (7.05, 6.9, 2.9, -0.5),
             (5.0, 5.0, 2.0, -0.5), (5.2, 5.0, 3.05, -0.6), (5.0, 5.3, 2.0, -0.7), (9.4, 9.5, 2.0, -2.4),
             (5.0, 5.0, 2.0, -2.4), (5.2, 5.2, 2.1, -2.2), (5.4, 5.4, 2.2, -2.0), (5.6, 5.6, 2.3, -1.8)]
    out = dict(pose_in=[], rot_yx=[], n=[], cloud_mm=[], cloud_off=[0], pos_after=[], count=[], life=[], state=[],
               idx_crc=[], loc_crc=[])
    for k, pose in enumerate(poses):
        n = int(rng.integers(800, 6000))
        cloud = synthetic_cloud(rng, n) * (1.6 if k in (3, 11) else 1.0)
        cloud = np.round(cloud * 1000) / 1000.0
        rot_yx = rng.uniform(-0.08, 0.08, 2)
        r.ref_set_cloud(vp(cloud), n)
        r.ref_set_scan_pose(*[float(v) for v in pose])
        r.ref_set_rot_yx(float(rot_yx[0]), float(rot_yx[1]))
        r.ref_scan_localization_map_update()
        st = dvec(r, "ref_get_scan_state", 18)
        out["pose_in"].append(pose); out["rot_yx"].append(rot_yx); out["n"].append(n)
        out["cloud_mm"].append(mm16(cloud)); out["cloud_off"].append(out["cloud_off"][-1] + n)
        out["pos_after"].append(st[0:3].copy()); out["count"].append(cnt_arr.copy()); out["life"].append(life_arr.copy())
        out["state"].append(dvec(r, "ref_get_ring_state", 7)); out["idx_crc"].append(crc(idx_arr))
        out["loc_crc"].append(crc(loc_grid(r)))
        print(f"ring call {k}: n={n} counts {cnt_arr.tolist()} life {life_arr.tolist()} pos after {st[0:3]}")
    np.savez_compressed(os.path.join(OUT, "ring.npz"), pose_in=np.array(out["pose_in"]), rot_yx=np.array(out["rot_yx"]),
                        n=np.array(out["n"], np.int32), cloud_mm=np.concatenate(out["cloud_mm"]),
                        cloud_off=np.array(out["cloud_off"], np.int64), pos_after=np.array(out["pos_after"]),
                        count=np.array(out["count"], np.int32), life=np.array(out["life"], np.int32),
                        state=np.array(out["state"]), idx_crc=np.array(out["idx_crc"], np.uint32),
                        loc_crc=np.array(out["loc_crc"], np.uint32), final_idx=idx_arr.copy())


def gen_frames(r, rng, frames=48, per_step=8, P=50):
    """The reference's own loop without ROS: per depth frame the calls of callback() in its order
    (S3P:173-191: resize_depth_to_array, img_to_cloud, IMU angles, scan_localization,
    scan_localization_map_update, build_local_map), and particle_map() (S3P:1106-1108) after every
    `per_step` frames. One sensor moving through the synthetic room; ends past a recentring."""
    r.ref_slam_reset()
    r.ref_alloc_particles(P)
    r.ref_init_maps()
    for i in range(P):
        r.ref_set_particle(i, 7.5, 7.5, 1.5, 0.0, 1.0 / P)
    r.ref_set_policy(1, 7.5, 7.5, 1.5, 0.0, 0)           # add_points = 1, pos_last_update (S3P:823-825)
    truth = np.array([5.0, 5.0, 2.0, 0.0])
    rec = dict(pooled=[], gyr=[], seed_vote=[], seed_rays=[], n=[], pick_crc=[], rays_crc=[], scan=[],
               ring_count=[], ring_life=[], ring_state=[], ring_idx_crc=[], loc_crc=[], local_crc=[],
               local_pose=[], gpose=[], cloud_crc=[])
    steps = dict(frame=[], seed=[], w_pre_crc=[], state=[], best=[], add_after=[], feedback=[], nbeams=[],
                 map_best_crc=[])
    idx_arr, cnt_arr, life_arr = ring_arrays(r)
    buf = C.create_string_buffer(1 << 22)
    seed = 300000
    for f in range(frames):
        step = np.array([0.055, 0.012 * np.cos(0.1 * f), 0.004, 0.009 * np.cos(0.08 * f)])
        truth = truth + step
        rot_y = 0.02 * np.sin(0.3 * f); rot_x = -0.015 * np.cos(0.2 * f)
        pooled = pooled_depth(truth[:3], (truth[3], rot_y, rot_x), f)
        raw = np.frombuffer(full_image(pooled).astype("<u2").tobytes(), np.uint8)
        n = r.ref_depth_to_cloud(vp(raw), len(raw))                          # S3P:173-175
        cloud = np.empty((n, 3)); r.ref_get_cloud(vp(cloud), n)
        gyr = (truth[3] + 0.002 * np.sin(f), rot_y, rot_x)
        r.ref_set_gyr(*[float(v) for v in gyr])                              # what sync_filter() leaves (S3P:244-246)
        seed, pick = screened_seed(r, seed + 1, 500, n)
        s_vote = seed
        r.ref_set_seed(seed)
        r.ref_scan_localization()                                            # S3P:187
        r.ref_scan_localization_map_update()                                 # S3P:189
        seed, rays = screened_seed(r, seed + 1, 2400, n)
        r.ref_set_seed(seed)
        r.ref_build_local_map()                                              # S3P:191
        rec["pooled"].append(pooled); rec["gyr"].append(gyr); rec["seed_vote"].append(s_vote)
        rec["seed_rays"].append(seed); rec["n"].append(n); rec["pick_crc"].append(crc(pick))
        rec["rays_crc"].append(crc(rays)); rec["cloud_crc"].append(crc(cloud))
        rec["scan"].append(dvec(r, "ref_get_scan_state", 18))
        rec["ring_count"].append(cnt_arr.copy()); rec["ring_life"].append(life_arr.copy())
        rec["ring_state"].append(dvec(r, "ref_get_ring_state", 7)); rec["ring_idx_crc"].append(crc(idx_arr))
        rec["loc_crc"].append(crc(loc_grid(r))); rec["local_crc"].append(crc(local_grid(r)))
        rec["local_pose"].append(dvec(r, "ref_get_local_pose", 4)); rec["gpose"].append(dvec(r, "ref_get_global_pose", 4))
        st = rec["scan"][-1]
        if f % 4 == 0:
            print(f"  frame {f}: n={n} scan pose {st[0]:.3f} {st[1]:.3f} {st[2]:.3f} yaw {st[3]:.3f} "
                  f"truth {truth[0]:.3f} {truth[1]:.3f} {truth[2]:.3f} {truth[3]:.3f} counts {cnt_arr.tolist()} life {life_arr.tolist()}",
                  flush=True)
        if (f + 1) % per_step == 0:
            nb = int(np.count_nonzero(local_grid(r)))
            seed += 1
            r.ref_set_seed(seed)                                             # noise seed; thresholds take seed + 1
            r.ref_particle_map(buf, len(buf))                                # S3P:1106-1108
            lines = buf.value.decode().split("\n")
            vals = lines[1].split()
            w_pre = np.array([float(v) for v in vals[:-1]])
            state = np.empty((P, 5)); tmp = (C.c_double * 5)()
            for i in range(P):
                r.ref_get_particle(i, tmp); state[i] = tmp[:]
            add = C.c_int(); best = C.c_int(); last = (C.c_double * 4)()
            r.ref_get_policy(C.byref(add), last, C.byref(best))
            steps["frame"].append(f); steps["seed"].append(seed); steps["w_pre_crc"].append(crc(w_pre))
            steps["state"].append(state); steps["best"].append(best.value); steps["add_after"].append(add.value)
            steps["feedback"].append(dvec(r, "ref_get_feedback", 4)); steps["nbeams"].append(nb)
            pm = np.ctypeslib.as_array(r.ref_particle_map_ptr(best.value), shape=(300 * 300 * 80,))
            steps["map_best_crc"].append(crc(pm))
            seed += 1
            print(f"  particle step after frame {f}: beams={nb} best={best.value} add={add.value} "
                  f"pose {state[best.value][:4]} global {rec['gpose'][-1]}", flush=True)
    recentred = [i for i in range(1, frames) if rec["scan"][i][0] == 5.0 and rec["scan"][i - 1][0] != 5.0]
    print("recentred at frames", recentred)
    final_local = local_grid(r).copy()
    np.savez_compressed(
        os.path.join(OUT, "frames.npz"), frames=np.int32(frames), per_step=np.int32(per_step), P=np.int32(P),
        pooled=np.array(rec["pooled"]), gyr=np.array(rec["gyr"]), seed_vote=np.array(rec["seed_vote"], np.int64),
        seed_rays=np.array(rec["seed_rays"], np.int64), n=np.array(rec["n"], np.int32),
        pick_crc=np.array(rec["pick_crc"], np.uint32), rays_crc=np.array(rec["rays_crc"], np.uint32),
        cloud_crc=np.array(rec["cloud_crc"], np.uint32), scan=np.array(rec["scan"]),
        ring_count=np.array(rec["ring_count"], np.int32), ring_life=np.array(rec["ring_life"], np.int32),
        ring_state=np.array(rec["ring_state"]), ring_idx_crc=np.array(rec["ring_idx_crc"], np.uint32),
        loc_crc=np.array(rec["loc_crc"], np.uint32), local_crc=np.array(rec["local_crc"], np.uint32),
        local_pose=np.array(rec["local_pose"]), gpose=np.array(rec["gpose"]),
        step_frame=np.array(steps["frame"], np.int32), step_seed=np.array(steps["seed"], np.int64),
        step_w_pre_crc=np.array(steps["w_pre_crc"], np.uint32), step_state=np.array(steps["state"]),
        step_best=np.array(steps["best"], np.int32), step_add_after=np.array(steps["add_after"], np.int32),
        step_feedback=np.array(steps["feedback"]), step_nbeams=np.array(steps["nbeams"], np.int32),
        step_map_best_crc=np.array(steps["map_best_crc"], np.uint32),
        final_local_idx=np.flatnonzero(final_local).astype(np.int32), final_local_val=final_local[final_local != 0],
        recentred=np.array(recentred, np.int32))


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/libs3p_ref.so missing: run oracle/build_ref.sh (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    r = load_ref()
    if len(sys.argv) > 1 and sys.argv[1] == "ties":         # the later additions, on their own: nothing else is rewritten
        gen_local_map_ties(r)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "vote_trig":
        gen_vote_trig(r)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "local_map_trig":
        gen_local_map_trig(r)
        return
    rng = np.random.default_rng(20261112)
    gen_helpers_p(r, rng)
    gen_depth(r, rng)
    gen_vote(r, rng)
    gen_local_map(r, rng)
    gen_ring(r, rng)
    gen_frames(r, rng)
    gen_local_map_ties(r)
    gen_vote_trig(r)
    gen_local_map_trig(r)
    print("front-end fixtures written to", OUT)


if __name__ == "__main__":
    main()
