"""GPU parity tests (-m gpu), per-frame half of the path: the HIP kernels behind pf_depth_to_cloud,
pf_scan_vote, pf_ring_*, pf_update_map and pf_read_occupied, called through the C ABI, against the
vectors produced by the REFERENCE'S OWN class Slam (tests/golden/{depth,vote,local_map,ring,frames,
helpers_p}.npz; generator oracle/make_golden_frontend.py). Bar: bit-exact."""
import numpy as np
import pytest

from slam_amd import capi
import golden_util as G

pytestmark = pytest.mark.gpu


def loc_handle():
    """The localisation grid (S3P:64: 200 x 200 x 80 binary uint8) with its keyframe ring."""
    h = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=200, grid_y=200,
                                        grid_z=80, local_x=0, local_y=0, local_z=0, max_beams=64))
    h.fill_maps(0)
    h.ring_init()
    return h


def mapper_handle(P=1):
    """The filter's handle at the reference's sizes: per-particle 300 x 300 x 80 maps + the int8
    local map build_local_map() writes (S3P:65)."""
    return capi.Handle(capi.default_config(n_particles=P, max_beams=1 << 18))


def test_depth_front_end_matches_reference(golden_dir):
    g = G.load(golden_dir, "depth.npz")
    h = loc_handle()
    try:
        cloud = h.depth_to_cloud(g["image"].astype("<u2").tobytes(), 640, 480)
        assert len(cloud) == int(g["n"])
        assert np.array_equal(cloud, g["cloud_mm"].astype(np.float64) / 1000.0)
    finally:
        h.close()


def test_depth_front_end_at_other_image_sizes():
    """The compaction's two forms (one workgroup per 256 pooled pixels up to 262 144 of them, one workgroup for all
    beyond) against the oracle's loop: a 700 x 420 image unpooled (294 000 pixels), the same pooled by 2, a
    one-pixel image, an all-zero image."""
    from oracle import binding as O
    rng = np.random.default_rng(3)
    h = loc_handle()
    try:
        for width, height, pool, zero in ((700, 420, 1, 0.4), (700, 420, 2, 0.4), (1, 1, 1, 0.0), (64, 48, 4, 1.0),
                                          (640, 480, 4, 0.97)):
            img = rng.integers(300, 9000, (height, width)).astype(np.uint16)
            img[rng.random((height, width)) < zero] = 0
            raw = np.frombuffer(img.astype("<u2").tobytes(), np.uint8)
            want = O.depth_to_cloud(raw, width, height, pool, 385.0 / pool)
            got = h.depth_to_cloud(raw, width, height, pool, 385.0 / pool)
            assert got.shape == want.shape and np.array_equal(got, want), (width, height, pool)
    finally:
        h.close()


def test_scan_vote_matches_reference(golden_dir):
    G.check_vote_cases(golden_dir, loc_handle)


def test_scan_vote_epilogue_trig_matches_reference(golden_dir):
    """scan_localization() of the reference at scan yaws where glibc's sincos() and sin()/cos() disagree in the
    last place (tests/golden/vote_trig.npz, `oracle/make_golden_frontend.py vote_trig`): pins the epilogue's
    sines and cosines (S3P:379-380: one sincos(yaw) for the first line, a separate sin(-yaw) for the second, as
    g++ -O2 compiles it and slam_amd/csrc/pf_host.cpp writes it) to the reference build itself."""
    G.check_vote_cases(golden_dir, loc_handle, "vote_trig.npz", 6)


def test_local_map_update_matches_reference(golden_dir):
    G.check_local_map_cases(golden_dir, mapper_handle)


def test_dead_reckoned_pose_trig_matches_reference(golden_dir):
    """build_local_map() of the reference with the GLOBAL yaw at angles where glibc's sincos() and sin()/cos()
    disagree (tests/golden/local_map_trig.npz, `oracle/make_golden_frontend.py local_map_trig`): pins
    pfh_update_realtime_pos (S3P:712-728, one sincos() for both lines in the g++ -O2 build) to the reference build;
    scan-pose increments of metres let the last place reach the dead-reckoned pose."""
    G.check_local_map_cases(golden_dir, mapper_handle, "local_map_trig.npz", 8)


def test_local_map_update_on_rounding_ties_matches_reference(golden_dir):
    """build_local_map() of the reference where its roundings sit on ties (tests/golden/local_map_ties.npz,
    made by `oracle/make_golden_frontend.py ties` from the reference's own class Slam): scan points at odd
    multiples of 25 mm along integer directions, at exactly 3 m, local yaws at which glibc's sincos() -- what
    g++ -O2 makes of the reference's sin() and cos() -- differs from the separate calls in the last place.
    Pins the choice of slam_amd/csrc/pf_trig.h to the reference build itself."""
    G.check_local_map_cases(golden_dir, mapper_handle, "local_map_ties.npz", 4)


def test_keyframe_ring_matches_reference(golden_dir):
    h = loc_handle()
    try:
        G.check_ring_calls(golden_dir, h)
    finally:
        h.close()


def test_read_out_matches_reference_index_split(golden_dir):
    """pf_read_occupied (S3P:1031-1051) on the reference's 300 x 300 x 80 grid: the cells of
    helpers_p.npz come back as the triples reverse_index_conv_p (S3P:100-110) gives for them."""
    g = G.load(golden_dir, "helpers_p.npz")
    js, rev = g["js_p"].astype(np.int64), g["rev_p"].astype(np.int64)
    h = mapper_handle()
    try:
        grid = np.zeros(300 * 300 * 80, np.uint8)
        grid[js] = 81
        h.write_map(0, grid)
        got, total = h.read_occupied(0, 80, 1)
        o = np.argsort(js); u = np.concatenate([[True], np.diff(js[o]) != 0])
        r = rev[o][u]
        want = np.stack([(r[:, 0] * 0.05), (-r[:, 2] * 0.05), (r[:, 1] * 0.05)], 1).astype(np.float32)
        assert total == len(want) and np.array_equal(got, want)
        sel = np.unique(js[js % 4 == 0])
        got4, total4 = h.read_occupied(0, 80, 4)
        assert total4 == len(sel)
    finally:
        h.close()


def test_callback_and_particle_loop_matches_reference(golden_dir):
    """The reference's own loop (48 callback() frames, particle_map() every 8, one recentring of the
    localisation grid) through the C ABI: depth image -> cloud -> vote -> ring -> local map ->
    snapshot -> measure/fuse -> weights -> resample, every member and map equal after every call."""
    loc, flt = loc_handle(), mapper_handle(50)
    try:
        G.check_frames(golden_dir, loc, flt, flt)
    finally:
        loc.close(); flt.close()


def test_callback_with_the_cloud_kept_on_the_gpu_matches_reference(golden_dir):
    """The same 48 frames and 6 filter steps through pf_depth_to_cloud_device, pf_scan_vote_device,
    pf_ring_update_device and pf_set_scan_device: the cloud (S3P:126) stays in one device buffer, the two handles
    share a stream, and every member, ring entry and map still equals the reference's after every call."""
    loc, flt = loc_handle(), mapper_handle(50)
    try:
        G.check_frames(golden_dir, loc, flt, flt, device_cloud=True)
    finally:
        flt.set_stream(None)                     # it borrowed loc's stream
        loc.close(); flt.close()


def test_device_cloud_forms_equal_the_host_forms_and_check_their_arguments(golden_dir):
    """Each pf_*_device call beside its host form on one cloud, and what the device forms do where the host
    forms would have refused: indices outside the cloud are refused all the same; a coordinate that is not
    finite or beyond 1e6 (nobody on the host sees it) makes its point contribute nothing."""
    import torch
    g = G.load(golden_dir, "local_map.npz")
    cloud = G.cloud_of(g["m0_fresh_cloud_mm"]); rays = g["m0_fresh_rays"]
    n = len(cloud)
    dev = torch.from_numpy(cloud.reshape(-1)).cuda()
    rng = np.random.default_rng(11)
    pos, zyx = np.array([5.0, 5.1, 1.6]), np.array([0.2, 0.01, -0.02])
    a, b = loc_handle(), loc_handle()
    try:
        grid = (rng.random(200 * 200 * 80) < 0.03).astype(np.uint8)
        a.write_map(0, grid); b.write_map(0, grid)
        pick = rng.integers(0, n, 500).astype(np.int32)
        sa, ba = a.scan_vote(cloud[pick], pos, zyx)
        sb, bb = b.scan_vote_device(dev.data_ptr(), n, pick, pos, zyx)
        assert np.array_equal(sa, sb) and ba == bb and sa.max() > 0
        s0, b0 = b.scan_vote_device(dev.data_ptr(), n, pick[:0], pos, zyx)
        assert not s0.any() and b0 == -1
        for bad in ([-1], [n]):
            with pytest.raises(capi.PfError):
                b.scan_vote_device(dev.data_ptr(), n, np.array(bad, np.int32), pos, zyx)
        fa, pa = a.ring_update(cloud, pos, zyx)
        fb, pb = b.ring_update_device(dev.data_ptr(), n, pos, zyx)
        assert fa == fb and np.array_equal(pa, pb)
        ra, rb = a.ring_state(with_idx=True), b.ring_state(with_idx=True)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)) and ra[0].sum() > 0
        with pytest.raises(capi.PfError):
            b.ring_update_device(dev.data_ptr(), 19201, pos, zyx)
    finally:
        a.close(); b.close()
    a, b = mapper_handle(), mapper_handle()
    try:
        params = capi.UpdateParams(10, 50, -3, -50, 3.0)
        a.set_scan(cloud, rays); a.update_map(capi.PF_GRID_LOCAL, params, pos, zyx)
        b.set_scan_device(dev.data_ptr(), n, rays); b.update_map(capi.PF_GRID_LOCAL, params, pos, zyx)
        want = a.read_local_map()
        assert np.array_equal(want, b.read_local_map()) and np.count_nonzero(want) > 1000
        for bad in ([-1], [n]):
            with pytest.raises(capi.PfError):
                b.set_scan_device(dev.data_ptr(), n, np.array(bad, np.int32))
        # wild coordinates: the points that hold them drop out, the others act as before
        wild = cloud.copy()
        rows = np.array([3, 700, 701, n - 1]); vals = [np.nan, np.inf, -2.0e6, 1.0e6]
        for r, v in zip(rows, vals):
            wild[r, r % 3] = v
        keep = np.setdiff1d(np.arange(n), rows)
        renum = np.full(n, -1, np.int64); renum[keep] = np.arange(len(keep))
        rays_kept = renum[rays[np.isin(rays, keep)]].astype(np.int32)
        a.reset_local_map(); b.reset_local_map()
        a.set_scan(cloud[keep], rays_kept); a.update_map(capi.PF_GRID_LOCAL, params, pos, zyx)
        wdev = torch.from_numpy(wild.reshape(-1)).cuda()
        b.set_scan_device(wdev.data_ptr(), n, rays); b.update_map(capi.PF_GRID_LOCAL, params, pos, zyx)
        assert np.array_equal(a.read_local_map(), b.read_local_map())
    finally:
        a.close(); b.close()


def test_scan_vote_leaves_the_scan_alone(golden_dir):
    """pf_set_scan -> pf_scan_vote -> pf_update_map on ONE handle: the vote's sampled points must not
    disturb the scan the update marches over."""
    from oracle import binding as O
    g = G.load(golden_dir, "local_map.npz")
    name = "m0_fresh"
    cloud = G.cloud_of(g[name + "_cloud_mm"])
    rays = g[name + "_rays"]
    dims = (200, 200, 80)
    h = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                                        grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=64,
                                        max_points=1 << 15))
    try:
        rng = np.random.default_rng(5)
        grid = (rng.random(dims) < 0.02).astype(np.uint8) * 40
        h.write_map(0, grid)
        pos, zyx = np.array([5.0, 5.1, 1.6]), np.array([0.2, 0.01, -0.02])
        params = capi.UpdateParams(10, 100, -3, 0, 3.0)
        h.set_scan(cloud, rays)
        h.scan_vote(cloud[rng.integers(0, len(cloud), 500)], pos, zyx)
        h.update_map(capi.PF_GRID_SHARED, params, pos, zyx)
        want = grid.copy()
        O.update_map(O.UpdateCfg(*dims, 0.05, 0, 10, 100, -3, 0, 3.0), want.reshape(-1), pos, zyx, cloud, rays)
        assert np.array_equal(h.read_map(0), want.reshape(-1))
    finally:
        h.close()


def test_random_front_end_cases():
    """tools/fuzz_front.py, fifty random cases: the scan vote with its lattice varied (1 to 16 yaws, 1 to 8
    translations per axis) through pf_scan_vote and pf_scan_vote_static, the keyframe ring over fourteen frames
    (slot sizes 50 to 3000, 1 to 8 slots, thresholds varied, replacements and recentrings), the depth front-end
    at random image shapes and pools, the read-out at random strides and thresholds and the int8 local-map
    update with every increment and clamp at the type's limits; coordinates snapped onto quarter voxels so
    that round() sits on ties. Everything exact against the oracle."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_front.py"), "50", "3"],
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-300:]
