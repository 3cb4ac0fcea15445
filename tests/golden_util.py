"""Shared by the CPU and GPU suites: readers of the front-end fixtures under tests/golden/ (made by
oracle/make_golden_frontend.py from the reference's own class Slam) and the checks both suites run
on a backend pair (oracle test doubles on the CPU, slam_amd.capi handles on the GPU)."""
import os
import zlib

import numpy as np

from slam_amd import capi
from slam_amd.filter import ParticleMapStep
from slam_amd.frontend import FrameLoop

NL = 200 * 200 * 80
PRE_TABLE = np.array([-50, -49, -48, -3, 0, 0, 0, 10, 41, 45, 50], np.int8)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def pre_map(k):
    """Input local map of local_map.npz case k (the same lines as oracle/make_golden_frontend.py)."""
    if k < 0:
        return np.zeros(NL, np.int8)
    j = np.arange(NL, dtype=np.int64)
    return PRE_TABLE[((j * 2654435761 + k * 97) >> 11) % len(PRE_TABLE)]


def cloud_of(mm):
    return mm.astype(np.float64) / 1000.0       # S3P:603-605


def case_names(g, suffix):
    return sorted({k[:-len(suffix)] for k in g.files if k.endswith(suffix)}, key=lambda s: int(s[1:s.index("_")]))


def full_image(pooled):
    return np.repeat(np.repeat(pooled, 4, 0), 4, 1).astype("<u2").tobytes()


# ---- scan_localization(): vote.npz ------------------------------------------------------------
def check_vote_cases(golden_dir, make_loc, fixture="vote.npz", at_least=10):
    """make_loc() -> a fresh localisation handle (grid all zero). Every case of the fixture: members
    before the call -> the members the reference's scan_localization() left (S3P:300-410)."""
    g = load(golden_dir, fixture)
    names = case_names(g, "_seed")
    assert len(names) >= at_least
    for name in names:
        loc = make_loc()
        grid = np.zeros(NL, np.uint8); grid[g[name + "_grid_idx"]] = 1
        loc.write_map(0, grid)
        fl = FrameLoop(loc, None, replay=True)
        before, after = g[name + "_before"], g[name + "_after"]
        cloud = cloud_of(g[name + "_cloud_mm"])
        fl.n = len(cloud); fl.cloud[:fl.n] = cloud
        fl.pos_scan[:] = before[0:3]; fl.yaw_scan = before[3]; fl.vote_loc[:] = before[4:8]
        fl.gyr_yaw_old = before[14]
        fl.ring_paint = None
        fl.set_gyr(g[name + "_gyr"])
        loc.ring_paint = lambda: None            # these cases set the grid directly: no ring
        pick = fl.scan_localization(int(g[name + "_seed"]))
        assert np.array_equal(pick, g[name + "_pick"]), name
        got = np.array([*fl.pos_scan, fl.yaw_scan, *fl.vote_loc, *fl.pos_diff_scan, fl.yaw_diff_scan, fl.yaw_diff])
        want = np.concatenate([after[0:13]])
        assert np.array_equal(got, want), (name, got, want)
        assert fl.gyr_yaw_old == after[14] and fl.rot_y == after[15] and fl.rot_x == after[16], name
        if hasattr(loc, "close"):
            loc.close()


# ---- build_local_map(): local_map.npz ---------------------------------------------------------
def check_local_map_cases(golden_dir, make_mapper, fixture="local_map.npz", at_least=5):
    g = load(golden_dir, fixture)
    names = case_names(g, "_seed")
    assert len(names) >= at_least
    for name in names:
        mapper = make_mapper()
        pre = pre_map(int(g[name + "_pre_k"]))
        mapper.write_local_map(pre)
        fl = FrameLoop(None, mapper, replay=True)
        cloud = cloud_of(g[name + "_cloud_mm"])
        fl.n = len(cloud); fl.cloud[:fl.n] = cloud
        lp = g[name + "_local_pose"]; gp = g[name + "_gpose"]
        fl.local_pos[:] = lp[:3]; fl.local_yaw = lp[3]
        fl.pos[:] = gp[:3]; fl.yaw = gp[3]
        fl.feedback[:] = g[name + "_fb"]
        d = g[name + "_diffs"]
        fl.pos_diff_scan[:] = d[:3]; fl.yaw_diff = d[3]
        fl.rot_y, fl.rot_x = g[name + "_rot_yx"]
        rays = fl.build_local_map(int(g[name + "_seed"]))
        assert np.array_equal(rays, g[name + "_rays"]), name
        post = mapper.read_local_map()
        touched = np.flatnonzero(post != pre).astype(np.int32)
        assert np.array_equal(touched, g[name + "_touched"]), (name, len(touched), len(g[name + "_touched"]))
        assert np.array_equal(post[touched], g[name + "_touched_val"]), name
        assert crc(post) == int(g[name + "_map_crc"]), name
        assert np.array_equal(np.array([*fl.local_pos, fl.local_yaw]), g[name + "_local_pose_after"]), name
        assert np.array_equal(np.array([*fl.pos, fl.yaw]), g[name + "_gpose_after"]), name
        assert np.array_equal(fl.feedback, g[name + "_fb_after"]), name
        if hasattr(mapper, "close"):
            mapper.close()


# ---- scan_localization_map_update(): ring.npz -------------------------------------------------
def check_ring_calls(golden_dir, loc):
    g = load(golden_dir, "ring.npz")
    off = g["cloud_off"]
    for k in range(len(g["n"])):
        cloud = cloud_of(g["cloud_mm"][off[k]:off[k + 1]])
        pose = g["pose_in"][k]
        flags, pos = loc.ring_update(cloud, pose[:3], (pose[3], g["rot_yx"][k][0], g["rot_yx"][k][1]))
        counts, lives, pu, eu, idx = loc.ring_state(with_idx=True)
        assert np.array_equal(pos, g["pos_after"][k]), k
        assert np.array_equal(counts, g["count"][k]) and np.array_equal(lives, g["life"][k]), (k, counts, lives)
        assert np.array_equal(np.concatenate([pu, eu]), g["state"][k][:6]), k
        assert crc(idx.astype(np.int32)) == int(g["idx_crc"][k]), k
        assert crc(loc.read_map(0)) == int(g["loc_crc"][k]), k
    assert np.array_equal(idx.reshape(-1), g["final_idx"])


# ---- the whole loop: frames.npz ---------------------------------------------------------------
def check_frames(golden_dir, loc, mapper, filter_backend=None, frames=None, device_cloud=False):
    """callback() per frame (S3P:168-205) and, with a filter backend, particle_map() after every
    per_step frames (S3P:1102-1111): every member the reference's run left after each call."""
    g = load(golden_dir, "frames.npz")
    F = int(g["frames"]) if frames is None else frames
    per_step, P = int(g["per_step"]), int(g["P"])
    fl = FrameLoop(loc, mapper, replay=True, device_cloud=device_cloud)
    st = ParticleMapStep(filter_backend, P) if filter_backend is not None else None
    if st is not None:
        st.set_state(np.full(P, 7.5), np.full(P, 7.5), np.full(P, 1.5), np.zeros(P), np.full(P, 1.0 / P))
    k = 0
    for f in range(F):
        fl.depth(full_image(g["pooled"][f]))
        cloud = fl._cloud_dev[:3 * fl.n].cpu().numpy() if device_cloud else fl.cloud[:fl.n]
        assert fl.n == int(g["n"][f]) and crc(cloud) == int(g["cloud_crc"][f]), f
        fl.set_gyr(g["gyr"][f])
        pick = fl.scan_localization(int(g["seed_vote"][f]))
        assert crc(pick) == int(g["pick_crc"][f]), f
        fl.map_update()
        rays = fl.build_local_map(int(g["seed_rays"][f]))
        assert crc(rays) == int(g["rays_crc"][f]), f
        s = g["scan"][f]
        got = np.array([*fl.pos_scan, fl.yaw_scan, *fl.vote_loc, *fl.pos_diff_scan, fl.yaw_diff_scan, fl.yaw_diff])
        assert np.array_equal(got, s[0:13]), (f, got, s[0:13])
        counts, lives, pu, eu, idx = loc.ring_state(with_idx=True)
        assert np.array_equal(counts, g["ring_count"][f]) and np.array_equal(lives, g["ring_life"][f]), f
        assert np.array_equal(np.concatenate([pu, eu]), g["ring_state"][f][:6]), f
        assert crc(idx.astype(np.int32)) == int(g["ring_idx_crc"][f]), f
        assert crc(loc.read_map(0)) == int(g["loc_crc"][f]), f
        assert crc(mapper.read_local_map()) == int(g["local_crc"][f]), f
        assert np.array_equal(np.array([*fl.local_pos, fl.local_yaw]), g["local_pose"][f]), f
        assert np.array_equal(np.array([*fl.pos, fl.yaw]), g["gpose"][f]), f
        if st is not None and (f + 1) % per_step == 0:
            assert int(g["step_frame"][k]) == f
            beams, local_pose, global_pose = fl.snapshot()
            assert len(beams[0]) == int(g["step_nbeams"][k]), k
            seed = int(g["step_seed"][k])
            st.step(beams, local_pose, global_pose, capi.draw_noise(seed, P), capi.draw_uniform(seed + 1, P))
            fl.feedback[:] = st.feedback                                           # S3P:956-959
            w, _ = filter_backend.get_weights()
            state = np.stack([st.x, st.y, st.z, st.yaw, w], 1)
            assert np.array_equal(state, g["step_state"][k]), k
            assert st.best == int(g["step_best"][k]) and st.add_points == int(g["step_add_after"][k]), k
            assert np.array_equal(st.feedback, g["step_feedback"][k]), k
            assert crc(filter_backend.read_map(st.best)) == int(g["step_map_best_crc"][k]), k
            k += 1
    if F == int(g["frames"]) and st is None:
        final = mapper.read_local_map()
        idx = np.flatnonzero(final).astype(np.int32)
        assert np.array_equal(idx, g["final_local_idx"]) and np.array_equal(final[idx], g["final_local_val"])
    return fl
