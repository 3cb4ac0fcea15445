"""GPU parity tests (-m gpu): the HIP path through the C ABI against the oracle on the same
seeded inputs and against the vectors produced by the reference itself (tests/golden/).
Bar: bit-exact for every integer, index and (in reference order) fp64 quantity; log-weights
within 1e-5 relative, the tolerance BASELINE.json's north_star states."""
import os
import zlib

import numpy as np
import pytest

from oracle import binding as O
from slam_amd import capi, synth
from slam_amd.filter import ParticleMapStep

pytestmark = pytest.mark.gpu
ORIGIN = (100, 100, 30)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def local_from_sparse(idx, val, dims=(200, 200, 80)):
    local = np.zeros(dims[0] * dims[1] * dims[2], np.int8)
    local[idx] = val
    return local.reshape(dims)


def test_library_is_the_hip_build():
    L = capi.lib()
    assert os.path.samefile(capi.LIB_PATH, os.path.join(os.path.dirname(capi.__file__), "csrc",
                                                        "libslam_pf.so"))
    assert all(hasattr(L, s) for s in capi.SYMBOLS)


@pytest.mark.parametrize("name", ["step_P50_trig", "step_P50_add1", "step_P50_add0", "step_P100_add1",
                                  "step_P100_add0"])
def test_reference_single_step(golden_dir, name):
    """One particle_map() call of the reference, reference sizes, per-particle maps."""
    g = load(golden_dir, name + ".npz")
    P = int(g["P"])
    world = g["world"].reshape(-1)
    local = local_from_sparse(g["local_idx"], g["local_val"])
    beams = capi.snapshot_beams(local, ORIGIN)
    h = capi.Handle(capi.default_config(n_particles=P))
    try:
        h.broadcast_map(world)
        st = ParticleMapStep(h, P)
        s_in = g["state_in"]
        st.set_state(s_in[:, 0], s_in[:, 1], s_in[:, 2], s_in[:, 3], s_in[:, 4])
        st.add_points = int(g["add"]); st.last_update[:] = g["last_in"]
        # pre-resample weights first (bit-exact against the reference's cout stream)
        h.set_beams(*beams)
        # voxel indices of particle 0 against the oracle
        vi = h.voxel_indices(0)
        ov = O.voxel_indices(O.ref_grid(), *s_in[0, :4], beams[0], beams[1], beams[2])
        assert np.array_equal(vi, ov)
        slots = st.step(beams, g["local_pose"], g["global_pose"], g["noise"], g["u"])
        out = g["state_out"]
        w, logw = h.get_weights()
        px, py, pz, pyaw = h.get_poses()
        assert np.array_equal(px, out[:, 0]) and np.array_equal(py, out[:, 1])
        assert np.array_equal(pz, out[:, 2]) and np.array_equal(pyaw, out[:, 3])
        assert np.array_equal(st.x, out[:, 0]) and np.array_equal(st.yaw, out[:, 3])
        assert np.array_equal(w, out[:, 4])
        assert st.best == int(g["best"]) and st.last_nohit == 0 and st.last_ambiguous == 0
        assert st.add_points == int(g["add_after"])
        assert np.array_equal(st.feedback, g["feedback"])
        # log-weights (permuted with the particles) within 1e-5 relative of log(reference weight)
        ok = out[:, 4] > 0
        assert np.allclose(logw[ok], np.log(out[ok, 4]), rtol=1e-5, atol=1e-6)
        crc = np.array([zlib.crc32(h.read_map(i).tobytes()) for i in range(P)], np.uint32)
        assert np.array_equal(crc, g["map_crc"])
        for i in (0, 1, P - 1):
            m = h.read_map(i)
            idx = np.flatnonzero(m != world).astype(np.int32)
            assert np.array_equal(idx, g[f"map{i}_idx"]) and np.array_equal(m[idx], g[f"map{i}_val"])
        assert slots is not None and len(slots) == P
    finally:
        h.close()


def test_reference_200_step_run(golden_dir):
    """BASELINE.json configs[0] on the GPU: state CRC equal to the reference's at every step."""
    g = load(golden_dir, "run_P100_200steps.npz")
    P, steps = int(g["P"]), int(g["steps"])
    h = capi.Handle(capi.default_config(n_particles=P))
    try:
        h.broadcast_map(g["world"].reshape(-1))
        st = ParticleMapStep(h, P)
        st.set_state(np.full(P, 7.5), np.full(P, 7.5), np.full(P, 1.5), np.zeros(P),
                     np.full(P, 1.0 / P))
        off = g["local_off"]
        for t in range(steps):
            local = local_from_sparse(g["local_idx"][off[t]:off[t + 1]],
                                      g["local_val"][off[t]:off[t + 1]])
            beams = capi.snapshot_beams(local, ORIGIN)
            st.step(beams, g["local_pose"][t], g["global_pose"][t], g["noise"][t], g["u"][t])
            assert st.best == int(g["best"][t]), t
            assert np.array_equal(st.feedback, g["feedback"][t]), t
            w, _ = h.get_weights()
            state = np.ascontiguousarray(np.stack([st.x, st.y, st.z, st.yaw, w], 1))
            assert zlib.crc32(state.tobytes()) == int(g["state_crc"][t]), t
        px, py, pz, pyaw = h.get_poses()
        w, _ = h.get_weights()
        assert np.array_equal(np.stack([px, py, pz, pyaw, w], 1), g["state_final"])
        crc = np.array([zlib.crc32(h.read_map(i).tobytes()) for i in range(P)], np.uint32)
        assert np.array_equal(crc, g["map_crc"])
    finally:
        h.close()


def _shared_cfg(P, dims, **kw):
    return capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0],
                               grid_y=dims[1], grid_z=dims[2], local_x=0, local_y=0, local_z=0,
                               max_points=1 << 15, **kw)


def _oracle_step(grid, sc, w0, u, best0=0):
    g = O.ref_grid(*grid.shape)
    s, c = O.measure(g, sc["x"], sc["y"], sc["z"], sc["yaw"], grid.reshape(-1), 0, *sc["beams"], 0)
    w = w0.copy()
    _, logw = O.weights(w, s, int(c[-1]))
    w_pre = w.copy()
    d, sa, best, nohit = O.resample(w, u, best0)
    return s, c, w_pre, logw, sa, best, nohit, w


@pytest.mark.parametrize("P,M,dims,order", [
    (1, 64, (64, 64, 32), capi.PF_ORDER_EXACT),
    (63, 100, (64, 64, 32), capi.PF_ORDER_EXACT),
    (2000, 1500, (128, 128, 128), capi.PF_ORDER_AUTO),
    (5000, 777, (128, 128, 64), capi.PF_ORDER_EXACT),
])
def test_shared_grid_step_exact_order(P, M, dims, order):
    sc = synth.scene(dims, P, M, seed=P)
    rng = np.random.default_rng(P)
    w0 = rng.random(P) + 0.05; w0 /= w0.sum()
    u = capi.draw_uniform(1234 + P, P)
    s, c, w_pre, logw, sa, best, nohit, w_after = _oracle_step(sc["grid"], sc, w0, u)
    h = capi.Handle(_shared_cfg(P, dims, order_mode=order))
    try:
        h.write_map(0, sc["grid"])
        h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_weights(w0)
        h.set_beams(*sc["beams"])
        h.measure(0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        for i in (0, P // 2, P - 1):
            ov = O.voxel_indices(O.ref_grid(*dims), sc["x"][i], sc["y"][i], sc["z"][i],
                                 sc["yaw"][i], *sc["beams"][:3])
            assert np.array_equal(h.voxel_indices(i), ov)
        h.weights()
        gw, glw = h.get_weights()
        assert np.array_equal(gw, w_pre)                       # reference order: bit-exact
        ok = w_pre > 0
        assert np.allclose(glw[ok], logw[ok], rtol=1e-5, atol=1e-6)
        slots, gbest, gnohit, amb = h.resample(u)
        assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit and amb == 0
        gw2, _ = h.get_weights()
        assert np.array_equal(gw2, w_after)
        px, py, pz, pyaw = h.get_poses()
        assert np.array_equal(px, sc["x"][sa]) and np.array_equal(pyaw, sc["yaw"][sa])
    finally:
        h.close()


@pytest.mark.parametrize("P", [9000, 20000])
def test_shared_grid_fast_order_large(P):
    """FAST order at a size where the sequential chain would dominate: scores and resampled
    indices bit-exact, weights within fp64 rounding, log-weights within 1e-5 relative.
    """
    M, dims = 2048, (128, 128, 128)
    sc = synth.scene(dims, P, M, seed=7)
    w0 = np.full(P, 1.0 / P)
    u = capi.draw_uniform(99, P)
    s, c, w_pre, logw, sa, best, nohit, w_after = _oracle_step(sc["grid"], sc, w0, u)
    h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
    try:
        h.write_map(0, sc["grid"])
        h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_weights(w0)
        h.set_beams(*sc["beams"])
        h.measure(0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        h.weights()
        gw, glw = h.get_weights()
        assert np.allclose(gw, w_pre, rtol=1e-12, atol=0)
        assert np.allclose(glw, logw, rtol=1e-5, atol=1e-6)
        slots, gbest, gnohit, amb = h.resample(u)
        assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit
        gw2, _ = h.get_weights()
        assert np.allclose(gw2, w_after, rtol=1e-12, atol=0)
        px, py, pz, pyaw = h.get_poses()
        assert np.array_equal(px, sc["x"][sa]) and np.array_equal(pyaw, sc["yaw"][sa])
    finally:
        h.close()


def test_fast_order_forms_are_bit_identical(monkeypatch):
    """The three launch forms of K3/K4 in FAST order (folded, chain, single workgroup) are the same
    arithmetic on the same 256-particle blocks: every output bit-identical, also when draws sit
    on CDF boundaries (the re-search path)."""
    P, M, dims = 7001, 256, (64, 64, 64)
    sc = synth.scene(dims, P, M, seed=5)
    rng = np.random.default_rng(5)
    w0 = rng.random(P) + 1e-3; w0 /= w0.sum()
    out = []
    # default = stages folded into their neighbours; CHAIN = every stage its own launch;
    # MID = one workgroup per chain
    for mid, chain in (("0", "0"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("SLAM_PF_MID", mid)
        monkeypatch.setenv("SLAM_PF_CHAIN", chain)
        h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
        try:
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0); h.weights()
            gw, glw = h.get_weights()
            # thresholds on and next to boundaries of the CDF the device will build
            cdf = np.cumsum(gw)
            u = capi.draw_uniform(77, P)
            u[:200] = cdf[100:300]
            u[200:400] = np.nextafter(cdf[300:500], 1.0)
            u[400] = 0.0; u[401] = 1.0; u[402] = np.nextafter(1.0, 0.0)
            slots, gbest, gnohit, amb = h.resample(u)
            gw2, glw2 = h.get_weights()
            out.append((gw, glw, slots, gbest, gnohit, amb, gw2, glw2, h.get_poses()))
        finally:
            h.close()
    a = out[0]
    assert a[5] > 0                                           # the re-search path did run
    for b in out[1:]:
        for x, y in zip(a[:8], b[:8]):
            assert np.array_equal(x, y)
        for x, y in zip(a[8], b[8]):
            assert np.array_equal(x, y)


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def test_workgroup_cdf_is_the_sequential_cdf_bit_for_bit():
    """The CDF that settles boundary draws at large particle counts is computed by a whole
    workgroup (integer scan inside each binade of the running sum) and must equal the reference's
    one-add-after-another loop S3P:763-767 in every bit: against numpy's cumsum (sequential fp64
    adds) and against the device's own one-lane chain, on normalised weights of several sizes and
    on inputs made of ties, zeros, binade crossings and values the scan hands back to the chain."""
    rng = np.random.default_rng(11)
    cases = {}
    for n in (1, 7, 64, 4097, 80000, 300001):
        w = rng.random(n) + 1e-3
        cases[f"normalised_{n}"] = w / w.sum()
    w = np.full(80000, 1.0 / 80000); cases["flat_80000"] = w
    # exact ties: multiples of 2^-54 and 2^-55 meet the ulp 2^-53 of sums in [0.5, 1) half-way
    k = rng.integers(1, 1 << 12, 50000).astype(np.float64)
    cases["ties"] = np.concatenate([[0.5], k * 2.0 ** -54, k[:5000] * 2.0 ** -55])
    cases["ties_odd_start"] = np.concatenate([[0.5 + 2.0 ** -53], np.full(30000, 2.0 ** -54)])
    z = np.zeros(30000); z[20000] = 0.25; z[20001:20100] = rng.random(99) * 1e-3; z[25000] = 0.75
    cases["leading_zeros_one_hot"] = z
    cases["all_zero"] = np.zeros(9000)
    cases["wide_range"] = 10.0 ** rng.uniform(-30, 3, 40000)                 # many binades, unnormalised
    cases["tiny"] = rng.random(20000) * 1e-300                               # below the scan's range
    bad = rng.random(20000) / 20000
    bad[5000] = -0.125; bad[9000] = np.inf; bad[9001] = -np.inf; bad[15000] = np.nan
    cases["negative_inf_nan"] = bad
    # an element the scan cannot take, as the LAST of a thread's eight (the first scan round starts
    # at 1024 behind the chain): the seven before it must not feel it
    for nm, v in (("stop_last_of_thread_neg", -1e-9), ("stop_last_of_thread_nan", np.nan),
                  ("stop_last_of_thread_big", 4.0)):
        a = (rng.random(30000) + 0.5) / 30000
        a[1024 + 8 * 100 + 7] = v; a[1024 + 8 * 777 + 5] = v
        cases[nm] = a
    # the same in a round without ties, which runs the plain-sum form and not the pair form:
    # weights that are integers of 2^-57 (the ulp of sums in [2^-5, 2^-4), reached near index 3000,
    # left near 6000); a first stop at 3500 pins where the next round starts, the second is then
    # the last of thread 50's eight
    for nm, v in (("stop_last_of_thread_plain_neg", -3.0 * 2.0 ** -57), ("stop_last_of_thread_plain_nan", np.nan)):
        a = rng.integers(1 << 40, 1 << 41, 30000).astype(np.float64) * 2.0 ** -57
        a[3500] = -5.0 * 2.0 ** -57
        a[3500 + 1 + 8 * 50 + 7] = v
        cases[nm] = a
    neg = rng.random(12000) / 12000; neg[100] = -3.0; neg[8000] = 5.0        # sum negative for a while
    cases["negative_sum"] = neg
    cases["denormals"] = np.concatenate([[5e-324] * 100, rng.random(5000) * 1e-310, [1.0], [5e-324] * 100])
    cases["huge_after_small"] = np.concatenate([rng.random(10000) * 1e-5, [1e200], rng.random(10000), [1e308, 1e308]])
    # the many-workgroup form at the size it is for: a million normalised weights (about ten binade crossings behind the
    # first segment), a flat million, ties spread over every segment, a segment boundary exactly on a binade crossing,
    # and elements the scan cannot take in the middle of otherwise clean segments
    w = rng.random(1000000) + 1e-3; cases["normalised_1000000"] = w / w.sum()
    cases["flat_1048576"] = np.full(1 << 20, 2.0 ** -20)
    k = rng.integers(1, 1 << 10, 200000).astype(np.float64)
    cases["ties_many_segments"] = np.concatenate([[0.5], k * 2.0 ** -54])
    cases["crossing_on_segment_boundary"] = np.concatenate([np.full(8192, 2.0 ** -14), np.full(8192, 2.0 ** -14), rng.random(30000) * 2.0 ** -16])
    a = (rng.random(100000) + 0.5) / 100000
    a[8192 * 3 + 17] = -1e-9; a[8192 * 5] = np.nan
    cases["bad_elements_inside_segments"] = a
    a = (rng.random(100000) + 0.5) / 100000; a[8192 * 4 + 5000] = 3.0
    cases["jump_inside_segment"] = a
    h = capi.Handle(_shared_cfg(64, (16, 16, 16)))
    try:
        for name, w in cases.items():
            with np.errstate(all="ignore"):
                ref = np.concatenate([[0.0], np.cumsum(w)])                  # sequential, index order
            par = h.sequential_cdf(w, how=1)
            assert np.array_equal(_bits(par), _bits(ref)) or _same_bits_nan(par, ref), name
            multi = h.sequential_cdf(w, how=2)                               # the many-workgroup form (8192-element segments)
            assert np.array_equal(_bits(multi), _bits(ref)) or _same_bits_nan(multi, ref), name + " (many workgroups)"
            if w.size <= 100000:
                chain = h.sequential_cdf(w, how=0)
                assert np.array_equal(_bits(chain), _bits(ref)) or _same_bits_nan(chain, ref), name
    finally:
        h.close()


def _same_bits_nan(a, b):
    """Equal bits wherever b is not NaN, NaN where it is (the payload of a NaN is not a result)."""
    nan = np.isnan(b)
    return np.array_equal(np.isnan(a), nan) and np.array_equal(_bits(a)[~nan], _bits(b)[~nan])


@pytest.mark.parametrize("P", [50000, 300000])
def test_boundary_draws_at_large_counts_match_the_sequential_loop(P):
    """FAST order beyond the size where k_first_c rebuilds the CDF itself (50 k: folded launches
    + k_cdf_exact_if; 300 k: the unfolded chain): thresholds on and next to CDF boundaries, the
    resampled slots must be the ones the oracle's sequential loop draws on the same weights."""
    M, dims = 64, (32, 32, 32)
    sc = synth.scene(dims, P, M, seed=9)
    rng = np.random.default_rng(P)
    w0 = rng.random(P) + 1e-3; w0 /= w0.sum()
    h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
    try:
        h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0); h.weights()
        gw, _ = h.get_weights()
        cdf = np.cumsum(gw)                                                  # the sequential CDF of the device's weights
        u = capi.draw_uniform(5, P)
        u[:2000] = cdf[1000:3000]                                            # on a boundary: no interval matches
        u[2000:4000] = np.nextafter(cdf[P // 2:P // 2 + 2000], 1.0)          # just above: the next particle
        u[4000:6000] = np.nextafter(cdf[P - 2500:P - 500], 0.0)              # just below
        u[6000] = 0.0; u[6001] = 1.0; u[6002] = np.nextafter(1.0, 0.0)
        d, sa, best, nohit = O.resample(gw.copy(), u, 0)
        slots, gbest, gnohit, amb = h.resample(u)
        assert amb > 0
        assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit
    finally:
        h.close()


def test_fresh_thresholds_copied_ahead_at_large_counts(monkeypatch):
    """From 262 144 particles on pf_resample sends a fresh threshold set to the device on a stream of its own (two
    slots in turn) instead of letting the search read it through the host mapping: four resamples in a row, each with
    its own weights and thresholds, the second and third without waiting for the one before (pf_resample_async, two
    outstanding), against a handle created with SLAM_PF_U_PREFETCH=0 that reads through the mapping. (The oracle's
    loop is quadratic in the particle count: test_boundary_draws_at_large_counts...[300000] above holds one resample
    through the copied-ahead path against it.)"""
    P, M, dims = 270000, 16, (32, 32, 32)
    sc = synth.scene(dims, P, M, seed=4)
    rng = np.random.default_rng(77)
    ws = [rng.random(P) + 1e-3 for _ in range(4)]
    ws = [w / w.sum() for w in ws]
    us = [capi.draw_uniform(20 + k, P) for k in range(4)]

    def run(prefetch):
        monkeypatch.setenv("SLAM_PF_U_PREFETCH", "1" if prefetch else "0")
        h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
        try:
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_beams(*sc["beams"])
            out = []
            h.set_weights(ws[0])
            out.append(h.resample(us[0])[:3])
            for k in (1, 2):                              # two in flight: both slots of the threshold buffers in use
                h.set_weights(ws[k])
                h.resample_async(us[k], want_slots=True)
            res = [h.resample_result() for _ in (1, 2)]
            h.synchronize()
            out += [r[:3] for r in res]
            h.set_weights(ws[3])
            out.append(h.resample(us[3])[:3])
            return out
        finally:
            h.close()

    got, want = run(True), run(False)
    for k in range(4):
        assert np.array_equal(got[k][0], want[k][0]) and got[k][1:] == want[k][1:], k
        assert len(np.unique(got[k][0])) > P // 3
    assert not np.array_equal(got[0][0], got[1][0])


def test_draws_between_the_two_ambiguity_bands():
    """K4's band around a CDF boundary (FAST order): a draw inside the first, cheap band (k + 64 roundings of
    sums <= cdf[k]) is tested again with the sequential sum's error bounded through its partial sums block by
    block (draw_is_ambiguous, about half the band on an even CDF). Draws at three quarters of the first band
    from a boundary must no longer send the step to the sequential-order CDF, draws at a quarter of it still
    must, and either way the resampled slots are the ones the oracle's sequential loop draws."""
    P, M, dims = 50000, 64, (32, 32, 32)
    sc = synth.scene(dims, P, M, seed=9)
    rng = np.random.default_rng(P + 1)
    w0 = rng.random(P) + 0.5; w0 /= w0.sum()
    for frac, flagged in ((0.75, False), (0.25, True)):
        h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
        try:
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0); h.weights()
            gw, _ = h.get_weights()
            cdf = np.concatenate([[0.0], np.cumsum(gw)])                   # cdf[k] = sum of the first k weights
            bound = 1.125 * 2.0 ** -53
            u = capi.draw_uniform(6, P)
            k1 = np.arange(P // 2, P // 2 + 1000); k2 = np.arange(P - 3000, P - 2000)
            loose = lambda k: bound * (k + 1 + 64.0) * cdf[k + 1]          # the band of the interval above boundary k
            u[:1000] = cdf[k1] + frac * loose(k1)                          # above boundary k1: lands in interval k1 + 1
            u[1000:2000] = cdf[k2] - frac * bound * (k2 + 64.0) * cdf[k2]  # below boundary k2: in interval k2
            assert np.all(u[:1000] < cdf[k1 + 1]) and np.all(u[1000:2000] > cdf[k2 - 1])
            d, sa, best, nohit = O.resample(gw.copy(), u, 0)
            slots, gbest, gnohit, amb = h.resample(u)
            assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit
            assert (amb >= 2000) if flagged else (amb <= 3), (frac, amb)
        finally:
            h.close()


def test_systematic_thresholds_same_kernel():
    """north_star's systematic resampling = the same kernels fed (u0+i)/N."""
    P, M, dims = 3000, 512, (64, 64, 64)
    sc = synth.scene(dims, P, M, seed=3)
    w0 = np.full(P, 1.0 / P)
    u = (0.37 + np.arange(P)) / P
    s, c, w_pre, logw, sa, best, nohit, w_after = _oracle_step(sc["grid"], sc, w0, u)
    h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_EXACT))
    try:
        h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0); h.weights()
        slots, gbest, gnohit, amb = h.resample(u)
        assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit
    finally:
        h.close()


@pytest.mark.parametrize("dims", [(32, 24, 16), (31, 23, 19)])
def test_edge_cases_bounds_aliasing_and_empty(dims):
    """dims not multiples of 4 exercise the padding of K1's gather layout (4 x 4 x 4 bricks)."""
    P = 130
    rng = np.random.default_rng(5)
    grid = rng.integers(0, 101, dims).astype(np.uint8)
    # poses near and beyond the grid faces: out-of-bounds flats and the flat-only aliasing of
    # S3P:882 (y = -1 or z = gz alias into a neighbouring row) must match the oracle
    x = rng.uniform(-0.3, dims[0] * 0.05 + 0.3, P); y = rng.uniform(-0.3, dims[1] * 0.05 + 0.3, P)
    z = rng.uniform(-0.2, dims[2] * 0.05 + 0.2, P); yaw = rng.uniform(-3.2, 3.2, P)
    M = 900
    bx = rng.integers(-20, 21, M).astype(np.int32); by = rng.integers(-20, 21, M).astype(np.int32)
    bz = rng.integers(-10, 11, M).astype(np.int32)
    bv = rng.choice(np.array([-50, -3, 1, 20, 21, 50], np.int8), M)
    og = O.ref_grid(*dims)
    s, c = O.measure(og, x, y, z, yaw, grid.reshape(-1), 0, bx, by, bz, bv, 0)
    h = capi.Handle(_shared_cfg(P, dims))
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_beams(bx, by, bz, bv); h.measure(0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        for i in range(0, P, 13):
            assert np.array_equal(h.voxel_indices(i),
                                  O.voxel_indices(og, x[i], y[i], z[i], yaw[i], bx, by, bz))
        # empty beam list: num_match = 0 -> the reference divides by zero (S3P:928): NaN weights,
        # no draw matches, every particle keeps its slot, best index unchanged
        h.set_beams(bx[:0], by[:0], bz[:0], bv[:0]); h.measure(0)
        gs, gc = h.get_scores()
        assert not gs.any() and not gc.any()
        h.weights()
        w, _ = h.get_weights()
        assert np.isnan(w).all()
        slots, best, nohit, amb = h.resample(capi.draw_uniform(1, P))
        assert np.array_equal(slots, np.arange(P)) and nohit == P and best == 0
    finally:
        h.close()


def test_zero_weight_particles_and_ragged_sizes():
    for P in (2, 65, 1025):
        rng = np.random.default_rng(P)
        w0 = rng.random(P); w0[rng.random(P) < 0.5] = 0.0
        if not w0.any():
            w0[0] = 1.0
        sc = synth.scene((48, 48, 32), P, 200, seed=P)
        u = capi.draw_uniform(P, P)
        s, c, w_pre, logw, sa, best, nohit, w_after = _oracle_step(sc["grid"], sc, w0, u)
        h = capi.Handle(_shared_cfg(P, (48, 48, 32), order_mode=capi.PF_ORDER_EXACT))
        try:
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0); h.weights()
            gw, glw = h.get_weights()
            assert np.array_equal(gw, w_pre)
            assert np.all(np.isneginf(glw[w_pre == 0]))
            slots, gbest, gnohit, amb = h.resample(u)
            assert np.array_equal(slots, sa) and gbest == best and gnohit == nohit
        finally:
            h.close()


def _update_case(signed, dims, seed):
    rng = np.random.default_rng(seed)
    M = 3000
    pos = np.array([dims[0] * 0.025, dims[1] * 0.025, dims[2] * 0.02])
    zyx = np.array([0.4, 0.05, -0.08])
    d = rng.normal(size=(M, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = d * rng.uniform(0.2, 4.5, (M, 1))
    pts[:5] = 0.0                                   # zero-length rays (norm 0 -> no steps)
    ray_idx = rng.integers(0, M, 2400).astype(np.int32)
    if signed:
        grid = rng.integers(-50, 51, dims).astype(np.int8)
        prm = (10, 50, -3, -50)
    else:
        grid = rng.integers(0, 101, dims).astype(np.uint8)
        prm = (10, 100, -3, 0)
    return grid, pos, zyx, pts, ray_idx, prm


def test_update_map_local_int8_matches_oracle():
    """S3P:628-704 on the reference's int8 local map, against the restatement on a scene of this test's own
    (the restatement is pinned to the reference's build_local_map() by tests/golden/local_map*.npz, and the
    library to those fixtures directly by tests/test_gpu_frontend.py)."""
    dims = (200, 200, 80)
    grid, pos, zyx, pts, ray_idx, prm = _update_case(True, dims, 11)
    cfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 1, prm[0], prm[1], prm[2], prm[3], 3.0)
    want = grid.copy()
    O.update_map(cfg, want, pos, zyx, pts, ray_idx)
    h = capi.Handle(capi.default_config(n_particles=1, grid_x=8, grid_y=8, grid_z=8))
    try:
        h.write_local_map(grid)
        h.set_scan(pts, ray_idx)
        h.update_map(capi.PF_GRID_LOCAL, capi.UpdateParams(*prm, 3.0), pos, zyx)
        got = h.read_local_map().reshape(dims)
        assert np.array_equal(got, want)
        assert np.count_nonzero(want != grid) > 1000
        h.reset_local_map()
        assert not h.read_local_map().any()
    finally:
        h.close()


def test_update_map_shared_uint8_matches_oracle():
    dims = (128, 128, 96)
    grid, pos, zyx, pts, ray_idx, prm = _update_case(False, dims, 12)
    cfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 0, prm[0], prm[1], prm[2], prm[3], 3.0)
    want = grid.copy()
    O.update_map(cfg, want, pos, zyx, pts, np.arange(len(pts), dtype=np.int32))
    h = capi.Handle(_shared_cfg(4, dims))
    try:
        h.write_map(0, grid)
        h.set_scan(pts, np.arange(len(pts), dtype=np.int32))      # north-star: every beam is cast
        h.update_map(capi.PF_GRID_SHARED, capi.UpdateParams(*prm, 3.0), pos, zyx)
        assert np.array_equal(h.read_map(0).reshape(dims), want)
    finally:
        h.close()


def test_update_map_rays_on_rounding_ties():
    """Scan points at whole numbers of half voxels from the sensor: the step count round(norm * conv)
    (S3P:676) and the range test norm > 3 (S3P:637) then sit on ties that the last place of the rotated
    point decides, and with it the last place of the rotation entries: the sensor yaw is one at which
    glibc's cos() and sincos() differ (slam_amd/csrc/pf_trig.h). Found by tools/fuzz_step.py."""
    dims = (96, 124, 80)
    rng = np.random.default_rng(135)
    n = 600
    az, el = rng.uniform(-np.pi, np.pi, n), rng.uniform(-0.5, 0.5, n)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    d[:200] = np.eye(3)[rng.integers(0, 3, 200)] * rng.choice([-1.0, 1.0], 200)[:, None]   # along the sensor axes too
    k = rng.integers(4, 61, n)
    k[::7] = 60                                                       # exactly max_range before rotation
    pts = d * ((k + rng.choice([0.0, 0.5], n)) * 0.05)[:, None]
    pts[::7] = d[::7] * 3.0
    pos = np.array([2.4, 3.1, 2.0]) + rng.normal(0, 0.02, 3)
    zyx = np.array([0.40005245828215963, 0.01, -0.02])
    ridx = np.arange(n, dtype=np.int32)
    prm = (10, 100, -3, 0)
    cfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 0, prm[0], prm[1], prm[2], prm[3], 3.0)
    grid = rng.integers(0, 101, dims).astype(np.uint8)
    want = grid.copy()
    O.update_map(cfg, want, pos, zyx, pts, ridx)
    h = capi.Handle(_shared_cfg(4, dims))
    try:
        h.write_map(0, grid)
        h.set_scan(pts, ridx)
        h.update_map(capi.PF_GRID_SHARED, capi.UpdateParams(*prm, 3.0), pos, zyx)
        assert np.array_equal(h.read_map(0).reshape(dims), want)
    finally:
        h.close()


@pytest.mark.parametrize("world", [1, 2])
def test_update_carried_by_the_weight_kernels_matches_oracle(world):
    """pf_update_map(PF_GRID_SHARED) only records the update; pf_weights launches it inside its own
    two kernels (one-GPU handle: k3_raw_m_u / k3_norm_c_u; shard of a larger job: k3_raw_u). The
    grid read afterwards, and the next measure on its gather-layout copy, must be the oracle's;
    a second update recorded before anybody picked up the first must not lose it."""
    import torch
    dims = (96, 96, 64)
    P = 3000
    sc = synth.scene(dims, P, 400, seed=77)
    grid = sc["grid"].copy()
    grid[grid > 100] = 100
    pts, ray_idx = sc["points"], sc["ray_idx"]
    tp = sc["true_pose"]
    pos = tp[:3].copy(); zyx = np.array([tp[3], 0.0, 0.0])
    prm = (10, 100, -3, 0)
    ocfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 0, prm[0], prm[1], prm[2], prm[3], 3.0)
    want = grid.copy()
    for _ in range(3):
        O.update_map(ocfg, want, pos, zyx, pts, ray_idx)
    cfg = _shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST)
    if world > 1:
        cfg.n_particles_global = P * world
        cfg.particle_offset = P
    h = capi.Handle(cfg)
    try:
        h.write_map(0, grid); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); h.set_beams(*sc["beams"])
        h.set_scan(pts, ray_idx)
        upd = capi.UpdateParams(*prm, 3.0)
        h.measure(0)
        h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)      # recorded ...
        h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)      # ... run on its own by this one, which is recorded
        if world > 1:
            s, c = h.get_scores()
            both = torch.tensor(np.stack([np.concatenate([s, s]), np.concatenate([c, c])], 1).astype(np.int32),
                                device="cuda")
            h.weights(both.data_ptr())                         # carried by k3_raw_u / k3_norm_c_u
        else:
            h.weights()                                        # carried by k3_raw_m_u / k3_norm_c_u
        h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)      # picked up by the next map reader
        h.measure(0)
        s2, c2 = h.get_scores()
        assert np.array_equal(h.read_map(0).reshape(dims), want)
        og = O.ref_grid(*dims)
        es, ec = O.measure(og, sc["x"], sc["y"], sc["z"], sc["yaw"], want.reshape(-1), 0, *sc["beams"], 0)
        assert np.array_equal(s2, es) and np.array_equal(c2, ec)
        if world == 1:
            # a carried update followed by the resample and the next measure
            h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)
            h.weights()
            h.resample(capi.draw_uniform(5, P))
            O.update_map(ocfg, want, pos, zyx, pts, ray_idx)
            assert np.array_equal(h.read_map(0).reshape(dims), want)
            h.measure(0)                                       # the gather-layout copy followed
            s3, c3 = h.get_scores()
            px, py, pz, pyaw = h.get_poses()
            es, ec = O.measure(og, px, py, pz, pyaw, want.reshape(-1), 0, *sc["beams"], 0)
            assert np.array_equal(s3, es) and np.array_equal(c3, ec)
    finally:
        h.close()


def test_measure_sees_map_updates_in_both_layouts():
    """K1 reads a gather-layout copy of the shared grid; K5 keeps it in step with the
    reference-order copy. Measure -> update -> measure -> upload -> measure against the oracle
    on the same sequence of grids (dims not multiples of 8)."""
    dims = (90, 70, 44)
    P, M = 700, 600
    sc = synth.scene(dims, P, M, seed=21)
    grid = sc["grid"].copy()
    og = O.ref_grid(*dims)
    ucfg = O.UpdateCfg(dims[0], dims[1], dims[2], 0.05, 0, 10, 100, -3, 0, 3.0)
    tp = sc["true_pose"]
    pos = tp[:3].copy(); zyx = np.array([tp[3], 0.02, -0.01])
    h = capi.Handle(_shared_cfg(P, dims))
    try:
        h.write_map(0, grid); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_beams(*sc["beams"]); h.set_scan(sc["points"], sc["ray_idx"])
        for rnd in range(3):
            h.measure(0)
            s, c = O.measure(og, sc["x"], sc["y"], sc["z"], sc["yaw"], grid.reshape(-1), 0, *sc["beams"], 0)
            gs, gc = h.get_scores()
            assert np.array_equal(gs, s) and np.array_equal(gc, c), rnd
            h.update_map(capi.PF_GRID_SHARED, capi.UpdateParams(10, 100, -3, 0, 3.0), pos, zyx)
            O.update_map(ucfg, grid, pos, zyx, sc["points"], sc["ray_idx"])
            assert np.array_equal(h.read_map(0).reshape(dims), grid)
            pos = pos + np.array([0.07, -0.04, 0.01])
        # a map written from outside replaces both copies
        grid = np.random.default_rng(3).integers(0, 101, dims).astype(np.uint8)
        h.write_map(0, grid); h.measure(0)
        s, c = O.measure(og, sc["x"], sc["y"], sc["z"], sc["yaw"], grid.reshape(-1), 0, *sc["beams"], 0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        h.fill_maps(37); h.measure(0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, 37 * gc)
    finally:
        h.close()


def test_full_size_properties_c2():
    """BASELINE configs[1] at full size (10k particles x 4k beams, 256^3 shared grid), through
    size-independent properties: a sample of particles against the oracle, linearity of the
    score in the beam list, and resampling invariants."""
    P, M, dims = 10000, 4096, (256, 256, 256)
    sc = synth.scene(dims, P, M, seed=0)
    h = capi.Handle(_shared_cfg(P, dims))
    try:
        h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        bx, by, bz, bv = sc["beams"]
        h.set_beams(bx, by, bz, bv); h.measure(0)
        s_all, c_all = h.get_scores()
        pick = np.arange(0, P, 97)
        og = O.ref_grid(*dims)
        s, c = O.measure(og, sc["x"][pick], sc["y"][pick], sc["z"][pick], sc["yaw"][pick],
                         sc["grid"].reshape(-1), 0, bx, by, bz, bv, 0)
        assert np.array_equal(s_all[pick], s) and np.array_equal(c_all[pick], c)
        # linearity: scores over two halves of the beam list add up to the score over the whole
        half = M // 2
        h.set_beams(bx[:half], by[:half], bz[:half], bv[:half]); h.measure(0)
        s1, c1 = h.get_scores()
        h.set_beams(bx[half:], by[half:], bz[half:], bv[half:]); h.measure(0)
        s2, c2 = h.get_scores()
        assert np.array_equal(s1 + s2, s_all) and np.array_equal(c1 + c2, c_all)
        h.set_beams(bx, by, bz, bv); h.measure(0); h.weights()
        w, logw = h.get_weights()
        assert abs(w.sum() - 1.0) < 1e-9
        assert np.allclose(np.exp(logw.astype(np.float64)), w, rtol=1e-5, atol=1e-12)
        u = capi.draw_uniform(3, P)
        slots, best, nohit, amb = h.resample(u)
        assert nohit == 0
        w2, _ = h.get_weights()
        assert np.array_equal(w2, w[slots])
        drawn = np.unique(slots)
        assert np.array_equal(slots[drawn], drawn)             # survivors keep their slot
        assert w2[best] == w2.max() and best == int(np.argmax(w2))
        # the oracle on the same normalised weights gives the same slots (sequential CDF)
        wc = w.copy()
        d, sa, obest, onohit = O.resample(wc, u, 0)
        assert np.array_equal(slots, sa) and obest == best
    finally:
        h.close()


def test_predict_device_equals_predict():
    """pf_predict_device hands over a pose block that already lives in HBM (the bench's path):
    same scores, weights and slots as the host upload of the same poses."""
    import torch
    P, M, dims = 3000, 600, (96, 96, 64)
    sc = synth.scene(dims, P, M, seed=21)
    u = capi.draw_uniform(77, P)
    res = []
    for mode in ("host", "device"):
        h = capi.Handle(_shared_cfg(P, dims))
        try:
            h.write_map(0, sc["grid"]); h.set_beams(*sc["beams"])
            if mode == "host":
                h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            else:
                src = capi.Handle(_shared_cfg(P, dims))
                src.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); src.synchronize()
                blk = src.poses_tensor().clone(); torch.cuda.synchronize()
                src.close()
                h.predict_device(blk.data_ptr())
            h.measure(0); s, c = h.get_scores(); h.weights(); w, lw = h.get_weights()
            slots, best, nohit, amb = h.resample(u)
            res.append((s, c, w, slots, best, nohit, np.stack(h.get_poses())))
        finally:
            h.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_per_particle_fuse_with_collisions_matches_oracle():
    """K1+K2 on per-particle maps at a size between the reference's and configs[3]: random
    beams (many collide after rotation + rounding), mixed signs, several steps; every map
    byte, score and slot against the oracle. The order-dependent clamp-add of S3P:892-904 is
    what this pins."""
    P, dims, M = 300, (64, 64, 32), 5000
    rng = np.random.default_rng(17)
    grid0 = rng.integers(0, 101, dims).astype(np.uint8)
    x = rng.uniform(1.2, 2.0, P); y = rng.uniform(1.2, 2.0, P); z = rng.uniform(0.6, 1.0, P)
    yaw = rng.uniform(-3.2, 3.2, P)
    lin = np.sort(rng.choice(41 * 41 * 21, M, replace=False))       # ascending flat order, distinct
    bx = (lin // (41 * 21) - 20).astype(np.int32); by = ((lin // 21) % 41 - 20).astype(np.int32)
    bz = (lin % 21 - 10).astype(np.int32)
    bv = rng.choice(np.array([-50, -9, -3, 3, 10, 21, 30, 50], np.int8), M)
    og = O.ref_grid(*dims)
    maps = np.tile(grid0.reshape(1, -1), (P, 1))
    w = np.full(P, 1.0 / P)
    cfg = capi.default_config(n_particles=P, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                              local_x=0, local_y=0, local_z=0)
    h = capi.Handle(cfg)
    try:
        h.broadcast_map(grid0); h.predict(x, y, z, yaw); h.set_weights(w)
        h.set_beams(bx, by, bz, bv)
        for step in range(3):
            add = 1 if step != 1 else 0
            s, c = O.measure(og, x, y, z, yaw, maps.reshape(-1), maps.shape[1], bx, by, bz, bv, add)
            h.measure(add)
            gs, gc = h.get_scores()
            assert np.array_equal(gs, s) and np.array_equal(gc, c), step
            O.weights(w, s, int(c[-1])); h.weights()
            gw, _ = h.get_weights()
            assert np.array_equal(gw, w), step
            u = capi.draw_uniform(500 + step, P)
            d, sa, best, nohit = O.resample(w, u, 0)
            slots, gbest, gnohit, amb = h.resample(u)
            assert np.array_equal(slots, sa) and gnohit == nohit, step
            O.permute(sa, maps)
            for arr in (x, y, z, yaw):
                O.permute(sa, arr)
            for i in range(0, P, 7):
                assert np.array_equal(h.read_map(i), maps[i]), (step, i)
        assert len(np.unique([zlib.crc32(maps[i].tobytes()) for i in range(P)])) > 1
    finally:
        h.close()


def test_sharded_backend_on_rccl_single_rank():
    """The multi-GPU code paths with the one rank this box has, both must equal the plain handle:
    "torch": slam_amd/sharded.py's zero-copy views of the library's device buffers,
    all_gather_into_tensor on the nccl = RCCL backend, gathered pointers into pf_weights /
    pf_resample; "native": pf_comm_init, the all-gathers enqueued by the library itself."""
    import torch
    import torch.distributed as dist
    from slam_amd.sharded import ShardedBackend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        P, M, dims = 4000, 700, (96, 96, 64)
        sc = synth.scene(dims, P, M, seed=31)
        u = capi.draw_uniform(8, P)
        out = []
        # "native": both all-gathers as one RCCL group inside pf_weights; "native-two": one per call
        # (SLAM_PF_TWO_GATHERS=1); "native-late": a predict between pf_weights and pf_resample, which
        # must make pf_resample gather the new pose blocks itself
        x2 = sc["x"] + 0.01
        for sharded in (False, "torch", "native", "native-two", False, "native-late"):
            late = sharded == "native-late" or (sharded is False and len(out) == 4)
            if sharded == "native-two":
                os.environ["SLAM_PF_TWO_GATHERS"] = "1"
            h = capi.Handle(_shared_cfg(P, dims))
            os.environ.pop("SLAM_PF_TWO_GATHERS", None)
            native = str(sharded).startswith("native")
            be = ShardedBackend(h, dist, native=native) if sharded else h
            if sharded:
                assert be.native == native
            try:
                h.write_map(0, sc["grid"]); be.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
                be.set_beams(*sc["beams"]); be.measure(0); be.weights()
                w, lw = h.get_weights()
                if late:
                    be.predict(x2, sc["y"], sc["z"], sc["yaw"])
                r = be.resample(u)
                out.append((w, lw, r[0], r[1], np.stack(h.get_poses())))
                if sharded:
                    bp = be.best_pose(r[1])
                    assert bp == tuple(np.stack(h.get_poses())[:, r[1]])
            finally:
                h.close()
        for other in out[1:4]:
            for a, b in zip(out[0], other):
                assert np.array_equal(a, b)
        for a, b in zip(out[4], out[5]):
            assert np.array_equal(a, b)
        assert not np.array_equal(out[0][4], out[4][4])         # the late predict did change the poses
        # a pose block handed over on the device (pf_predict_device leaves it in place on a one-GPU
        # handle) must reach both kinds of all-gather
        dev = []
        for sharded in (False, "torch", "native"):
            h = capi.Handle(_shared_cfg(P, dims))
            be = ShardedBackend(h, dist, native=sharded == "native") if sharded else h
            try:
                h.write_map(0, sc["grid"]); be.predict(x2, sc["y"], sc["z"], sc["yaw"]); h.synchronize()
                block = h.poses_tensor().clone(); torch.cuda.synchronize()
                be.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
                be.set_beams(*sc["beams"]); be.measure(0)
                h.predict_device(block.data_ptr())
                be.weights()
                r = be.resample(u)
                dev.append((r[0], r[1], np.stack(h.get_poses())))
            finally:
                h.close()
        for other in dev[1:]:
            for a, b in zip(dev[0], other):
                assert np.array_equal(a, b)
        assert np.array_equal(np.sort(dev[0][2][0]), np.sort(x2[np.unique(dev[0][0])].repeat(np.unique(dev[0][0], return_counts=True)[1])))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_sharded_handles_at_nonzero_offsets_equal_one_gpu(world):
    """Every rank's kernels of a `world`-rank job, run one after the other on this one GPU: the
    gathered scores and pose blocks are assembled by hand from a one-GPU run of the same
    particles and handed to pf_weights / pf_resample as a collective would (particle_offset > 0,
    ancestors in other ranks' blocks). Each shard must end with its slice of the one-GPU result."""
    import torch
    Pg, M, dims = 5000, 600, (96, 96, 64)
    P = Pg // world
    sc = synth.scene(dims, Pg, M, seed=41)
    rng = np.random.default_rng(41)
    w0 = rng.random(Pg) + 0.01; w0 /= w0.sum()
    u = capi.draw_uniform(9, Pg)
    full = capi.Handle(_shared_cfg(Pg, dims, order_mode=capi.PF_ORDER_FAST))
    try:
        full.write_map(0, sc["grid"]); full.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        full.set_weights(w0); full.set_beams(*sc["beams"]); full.measure(0)
        fs, fc = full.get_scores()
        full.synchronize()
        pose_full = full.poses_tensor().clone()                         # [9, Pg]
        full.weights()
        fw, flw = full.get_weights()
        fslots, fbest, fnohit, famb = full.resample(u)
        fw2, flw2 = full.get_weights()
        fposes = np.stack(full.get_poses())
    finally:
        full.close()
    scores_all = torch.from_numpy(np.stack([fs, fc], 1).astype(np.int32)).cuda().contiguous()
    poses_all = torch.cat([pose_full[:, r * P:(r + 1) * P].contiguous().reshape(-1) for r in range(world)]).contiguous()
    torch.cuda.synchronize()
    for r in range(world):
        lo = r * P
        h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST, n_particles_global=Pg,
                                    particle_offset=lo))
        try:
            h.write_map(0, sc["grid"])
            h.predict(sc["x"][lo:lo + P], sc["y"][lo:lo + P], sc["z"][lo:lo + P], sc["yaw"][lo:lo + P])
            h.set_weights(w0); h.set_beams(*sc["beams"]); h.measure(0)
            s, c = h.get_scores()
            assert np.array_equal(s, fs[lo:lo + P]) and np.array_equal(c, fc[lo:lo + P])
            h.weights(scores_all.data_ptr())
            w, lw = h.get_weights()
            assert np.array_equal(w, fw) and np.array_equal(lw, flw)      # same bits on every rank
            slots, best, nohit, amb = h.resample(u, poses_all.data_ptr())
            assert np.array_equal(slots, fslots) and (best, nohit, amb) == (fbest, fnohit, famb)
            w2, lw2 = h.get_weights()
            assert np.array_equal(w2, fw2) and np.array_equal(lw2, flw2)
            assert np.array_equal(np.stack(h.get_poses()), fposes[:, lo:lo + P])
        finally:
            h.close()


@pytest.mark.parametrize("seed,n_points", [(21, 500), (22, 37), (23, 0)])
def test_scan_vote_matches_oracle(seed, n_points):
    """S3P:328-376 (625 pose hypotheses x 500 points on the binary localisation grid).
    Against the restatement on scenes of this test's own; the restatement is pinned to the reference's
    scan_localization() by tests/golden/vote*.npz (tests/test_frontend_golden.py) and the library to those
    fixtures directly (tests/test_gpu_frontend.py). Integer-exact."""
    rng = np.random.default_rng(seed)
    dims = (200, 200, 80)
    world = synth.make_room(dims, seed=seed)
    grid = (world > 50).astype(np.uint8)               # S3P:697-699: map_localization is 0/1
    pos = np.array([5.0, 5.0, 1.5]) + rng.normal(0, 0.05, 3)
    zyx = np.array([rng.uniform(-3, 3), rng.normal(0, 0.02), rng.normal(0, 0.02)])
    # points near set cells, expressed in the sensor frame, plus far ones that fall outside
    occ = np.argwhere(grid > 0)
    pick = occ[rng.integers(0, len(occ), n_points)] if n_points else np.zeros((0, 3), np.int64)
    wp = pick * 0.05 + rng.normal(0, 0.02, (n_points, 3))
    R = np.array(O.rot_zyx(*zyx)).reshape(3, 3) if hasattr(O, "rot_zyx") else None
    if R is None:
        cz, sz = np.cos(zyx[0]), np.sin(zyx[0])
        R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    pts = (wp - pos) @ R                                # R^T (wp - pos) per row
    if n_points > 8:
        pts[:4] *= 40.0                                 # off the grid (flat-only bounds rule)
        pts[4] = [1e5, -1e5, 3e5]                       # far outside the grid
    want, want_best = O.scan_vote(grid, 0.05, pts, pos, zyx)
    h = capi.Handle(capi.default_config(n_particles=1, map_mode=capi.PF_MAP_SHARED,
                                        grid_x=dims[0], grid_y=dims[1], grid_z=dims[2],
                                        max_points=1024))
    try:
        h.write_map(0, grid.reshape(-1))
        got, got_best = h.scan_vote(pts, pos, zyx)
    finally:
        h.close()
    assert np.array_equal(got, want)
    assert got_best == want_best
    if n_points >= 100:
        assert want.max() > n_points // 4               # the case does vote
    if n_points == 0:
        assert got_best == -1


def _ring_frames(dims, n_frames, cap, seed):
    """A drifting sensor with scans of a room: enough motion for slot replacement every few
    frames and for the recentring (S3P:485) to trigger, some points outside the grid (holes)."""
    rng = np.random.default_rng(seed)
    res = 0.05
    room = synth.make_room(dims, seed=seed)
    occ = np.argwhere(room == 100) * res
    centre = np.array(dims) * res / 2
    pos = centre + np.array([0.3, -0.2, 0.05])
    yaw = 0.1
    frames = []
    for t in range(n_frames):
        pos = pos + np.array([0.11, 0.07, 0.012]) + rng.normal(0, 0.01, 3)
        yaw += 0.03
        zyx = np.array([yaw, rng.normal(0, 0.01), rng.normal(0, 0.01)])
        n = int(rng.integers(cap // 2, cap + 1)) if t % 7 else 0
        wp = occ[rng.integers(0, len(occ), n)] + rng.normal(0, 0.01, (n, 3))
        if n > 10:
            wp[::9] += 40.0                                   # far outside the grid
        cz, sz = np.cos(zyx[0]), np.sin(zyx[0])
        R = np.array(O.rot_zyx(*zyx)).reshape(3, 3)
        cloud = (wp - pos) @ R                                 # sensor frame
        frames.append((cloud, pos.copy(), zyx))
    return frames


def test_keyframe_ring_matches_oracle():
    """S3P:305-312, 419-529: slot replacement with the non-compacted write,
    painting, erasing, and the flat-offset recentring. Against the restatement (pinned to the
    reference's scan_localization_map_update() by tests/golden/ring.npz and frames.npz): every index of every slot
    after every frame, then a vote (K6) on the painted grid."""
    dims = (120, 100, 48)
    cap = 3000
    frames = _ring_frames(dims, 45, cap, 7)
    ring = O.RingState(dims, 0.05, n_slots=5, cap=cap)
    grid = np.zeros(dims, np.uint8)
    h = capi.Handle(_shared_cfg(1, dims))
    try:
        h.fill_maps(0)
        h.ring_init(slot_capacity=cap)
        seen = 0
        for t, (cloud, pos, zyx) in enumerate(frames):
            ring.paint(grid); h.ring_paint()
            assert np.array_equal(h.read_map(0).reshape(dims), grid), t
            of, opos = ring.update(grid, cloud, pos, zyx)
            gf, gpos = h.ring_update(cloud, pos, zyx)
            assert gf == of and np.array_equal(gpos, opos), t
            counts, lives, pu, eu, idx = h.ring_state(with_idx=True)
            assert np.array_equal(counts, ring.count) and np.array_equal(lives, ring.life), t
            assert np.array_equal(pu, np.array(ring.r.pos_update)) and np.array_equal(eu, np.array(ring.r.euler_update)), t
            assert np.array_equal(idx.reshape(-1), ring.idx), t
            assert np.array_equal(h.read_map(0).reshape(dims), grid), t
            seen |= of
            if of & 2:                                         # the caller carries the reset pose on
                for k in range(t + 1, len(frames)):
                    frames[k] = (frames[k][0], frames[k][1] - (pos - opos), frames[k][2])
        assert seen == 3                                       # both branches ran
        ring.paint(grid); h.ring_paint()
        cloud, pos, zyx = frames[-1]
        want, wbest = O.scan_vote(grid, 0.05, cloud[:500], pos, zyx)
        got, gbest = h.scan_vote(cloud[:500], pos, zyx)
        assert np.array_equal(got, want) and gbest == wbest and want.max() > 0
    finally:
        h.close()


@pytest.mark.parametrize("shape", [(640, 480, 4), (50, 31, 3), (64, 48, 1), (16, 16, 16)])
def test_depth_front_end_matches_oracle(shape):
    """S3P:562-612: pooled depth image -> sensor-frame cloud, bit-exact against
    the restatement (itself pinned by the reference's callback() body run here: tests/golden/depth.npz, tests/test_frontend_golden.py). Images with holes (zero
    depth), saturated pixels and fully empty pooled blocks; ragged sizes drop the remainder
    columns/rows like the reference's res_h = width/pool."""
    width, height, pool = shape
    rng = np.random.default_rng(width * 7 + pool)
    img = rng.integers(300, 9000, (height, width)).astype(np.uint16)
    img[rng.random((height, width)) < 0.35] = 0
    img[rng.random((height, width)) < 0.01] = 65535
    img[: height // 3, : width // 4] = 0                      # empty pooled blocks
    raw = np.frombuffer(img.astype("<u2").tobytes(), np.uint8)
    focal = 385.0 / pool
    want = O.depth_to_cloud(raw, width, height, pool, focal)
    h = capi.Handle(_shared_cfg(1, (16, 16, 16)))
    try:
        got = h.depth_to_cloud(raw, width, height, pool, focal)
        assert got.shape == want.shape and len(want) > 0
        assert np.array_equal(got, want)
        empty = h.depth_to_cloud(np.zeros(width * height * 2, np.uint8), width, height, pool, focal)
        assert empty.shape == (0, 3)
        with pytest.raises(capi.PfError):
            h.depth_to_cloud(raw, width, height, 0, focal)
    finally:
        h.close()


@pytest.mark.parametrize("dims,stride", [((200, 200, 80), 4), ((31, 23, 19), 4), ((64, 48, 40), 1),
                                         ((33, 17, 9), 7)])
def test_map_read_out_matches_oracle(dims, stride):
    """S3P:1031-1051: the published point cloud of the best particle's grid,
    bit-exact float32 triples in index order; truncation at the caller's capacity; per-particle
    slabs picked by global particle index."""
    rng = np.random.default_rng(dims[0] + stride)
    grid = rng.integers(0, 256, dims).astype(np.uint8)
    grid[rng.random(dims) < 0.7] = 50
    want = O.read_occupied(grid, 0.05, 80, stride)
    h = capi.Handle(_shared_cfg(1, dims))
    try:
        h.broadcast_map(grid)
        got, total = h.read_occupied(0, 80, stride)
        assert total == len(want) and got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        few, total2 = h.read_occupied(0, 80, stride, capacity=5)
        assert total2 == total and np.array_equal(few, want[:5])
        none, _ = h.read_occupied(0, 255, stride)
        assert none.shape == (0, 3)
        with pytest.raises(capi.PfError):
            h.read_occupied(0, 80, 0)
    finally:
        h.close()
    if dims[0] > 40:
        return
    P = 3
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_PER_PARTICLE, grid_x=dims[0], grid_y=dims[1],
                              grid_z=dims[2], max_beams=64, max_points=64)
    h = capi.Handle(cfg)
    try:
        grids = [np.roll(grid, 3 * i, axis=0) for i in range(P)]
        for i in range(P):
            h.write_map(i, grids[i])
        for i in range(P):
            got, _ = h.read_occupied(i, 80, stride)
            assert np.array_equal(got, O.read_occupied(grids[i], 0.05, 80, stride))
        with pytest.raises(capi.PfError):
            h.read_occupied(P, 80, stride)
    finally:
        h.close()


def test_sharded_per_particle_maps_need_the_library_communicator():
    """Per-particle maps on a shard of a larger job (FastSLAM mode): the slabs of remote
    ancestors cross ranks inside pf_resample (RCCL send/receive; the plan is tested on the CPU by
    tests/test_capi_cpu.py::test_migration_plan_moves_every_remote_ancestor, the transfers
    themselves need more than the one GPU of this box). Without pf_comm_init the call must fail
    with a clear message instead of leaving remote slots stale; measure, weights and every local
    accessor work on such a handle."""
    import torch
    P, world, dims = 64, 2, (24, 24, 16)
    sc = synth.scene(dims, P, 40, seed=5)
    cfg = capi.default_config(n_particles=P, n_particles_global=P * world, particle_offset=P,
                              map_mode=capi.PF_MAP_PER_PARTICLE, grid_x=dims[0], grid_y=dims[1],
                              grid_z=dims[2], max_beams=64, max_points=64)
    h = capi.Handle(cfg)
    try:
        h.broadcast_map(sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_beams(*sc["beams"]); h.measure(1)
        s, c = h.get_scores()
        scores = torch.tensor(np.stack([np.concatenate([s, s]), np.concatenate([c, c])], 1).astype(np.int32),
                              device="cuda")
        h.weights(scores.data_ptr())
        poses = h.poses_tensor()
        both = torch.cat([poses, poses], 0).contiguous()
        with pytest.raises(capi.PfError, match="pf_comm_init"):
            h.resample(capi.draw_uniform(3, P * world), both.data_ptr(), True)
    finally:
        h.close()


def test_async_resample_pipeline_equals_the_blocking_calls():
    """pf_resample_async / pf_resample_result: the results of step i read after step i+1 has been
    enqueued must be the ones the blocking pf_resample returns, step by step, the slots included;
    the final state is the same; the queue is two deep and says so."""
    import torch
    P, M, dims, steps = 6000, 500, (96, 96, 64), 6
    sc = synth.scene(dims, P, M, seed=61)
    u = capi.draw_uniform(17, P)
    # thresholds: new ones in three steps running (the pinned staging slots alternate and are reused
    # while a resample is in flight), then the last set again, a new one, the last set again
    us = [u, capi.draw_uniform(18, P), capi.draw_uniform(19, P), None, capi.draw_uniform(20, P), None]
    upd = capi.UpdateParams(10, 100, -3, 0, 3.0)
    tp = sc["true_pose"]
    pos = tp[:3].copy(); zyx = np.array([tp[3], 0.0, 0.0])
    rng = np.random.default_rng(61)
    hosts = [(tp[0] + rng.normal(0, 0.03, P), tp[1] + rng.normal(0, 0.03, P), tp[2] + rng.normal(0, 0.015, P),
              tp[3] + rng.normal(0, 0.02, P)) for _ in range(steps)]
    runs = []
    for mode in ("blocking", "async"):
        h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
        try:
            blocks = []
            for xs in hosts:                                   # pose blocks resident in HBM, as bench.py does
                h.predict(*xs); h.synchronize(); blocks.append(h.poses_tensor().clone())
            torch.cuda.synchronize()
            h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
            h.set_beams(*sc["beams"]); h.set_scan(sc["points"], sc["ray_idx"])
            res = []
            for i in range(steps):
                h.measure(0); h.update_map(capi.PF_GRID_SHARED, upd, pos, zyx)
                h.predict_device(blocks[i].data_ptr()); h.weights()
                if mode == "blocking":
                    res.append(h.resample(us[i]))
                else:
                    h.resample_async(us[i], None, True)
                    if i:
                        res.append(h.resample_result())
            if mode == "async":
                with pytest.raises(capi.PfError, match="outstanding|waits"):
                    h.resample(None)                           # one still in flight
                res.append(h.resample_result())
                with pytest.raises(capi.PfError, match="no resample outstanding"):
                    h.resample_result()
            runs.append((res, h.get_weights(), np.stack(h.get_poses()), h.read_map(0)))
        finally:
            h.close()
    (ra, wa, pa, ma), (rb, wb, pb, mb) = runs
    assert len(ra) == len(rb) == steps
    for a, b in zip(ra, rb):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1])
    assert np.array_equal(pa, pb) and np.array_equal(ma, mb)
    # the queue is two deep
    h = capi.Handle(_shared_cfg(P, dims, order_mode=capi.PF_ORDER_FAST))
    try:
        h.write_map(0, sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"]); h.set_beams(*sc["beams"])
        for i in range(2):
            h.measure(0); h.weights(); h.resample_async(u if i == 0 else None)
        h.measure(0); h.weights()
        with pytest.raises(capi.PfError, match="two resamples"):
            h.resample_async(None)
        h.resample_result(); h.resample_result()
    finally:
        h.close()


def test_measure_column_groups_of_every_shape():
    """K1 groups the occupied beams of one local column (k_measure_col, up to four per group, lz
    sorted inside a group, groups sorted by size): columns of 1 to 9 beams, lz in any order,
    repeated beams, columns that leave the grid in z for some particles and in x / y for others,
    beams below the occupancy threshold in between. Scores against the oracle."""
    dims = (48, 40, 24)
    P = 700
    rng = np.random.default_rng(91)
    grid = rng.integers(0, 101, dims).astype(np.uint8)
    x = rng.uniform(0.6, dims[0] * 0.05 - 0.6, P); y = rng.uniform(0.6, dims[1] * 0.05 - 0.6, P)
    z = rng.uniform(0.1, dims[2] * 0.05 - 0.1, P); yaw = rng.uniform(-3.2, 3.2, P)
    bx, by, bz, bv = [], [], [], []
    for _ in range(260):
        cx, cy = int(rng.integers(-14, 15)), int(rng.integers(-14, 15))
        n = int(rng.integers(1, 10))
        zs = rng.integers(-14, 15, n)                       # unsorted, may repeat
        for k, zz in enumerate(zs):
            bx.append(cx); by.append(cy); bz.append(int(zz))
            bv.append(50 if (k % 5) else 3)                 # every fifth beam is not occupied (v <= 20)
    bx = np.array(bx, np.int32); by = np.array(by, np.int32); bz = np.array(bz, np.int32)
    bv = np.array(bv, np.int8)
    og = O.ref_grid(*dims)
    s, c = O.measure(og, x, y, z, yaw, grid.reshape(-1), 0, bx, by, bz, bv, 0)
    h = capi.Handle(_shared_cfg(P, dims))
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_beams(bx, by, bz, bv); h.measure(0)
        gs, gc = h.get_scores()
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        assert c.min() < c.max()                            # some particles lose beams at the faces
    finally:
        h.close()


def test_random_whole_steps_on_the_shared_grid():
    """tools/fuzz_step.py, forty random scenes of three steps each (measure -> map update -> predict ->
    weights -> resample; particle counts beside every block size, ragged grids, beam lists of 1 to 5000,
    weights with zeros, one-hot and tiny): scores, weights, slots, best index, poses and the grid against
    the oracle after every step. Its scan points sit on half-voxel ranges, so the rays' step counts are on
    rounding ties throughout (the case that showed the sincos() difference is among these)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_step.py"), "40", "2"],
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-300:]


def test_random_sharded_jobs_equal_the_one_gpu_handle():
    """tools/fuzz_shard.py, twenty random jobs of 2 to 8 ranks run rank after rank on this GPU (shard sizes on
    and beside the block sizes, weights with zeros, one-hot and tiny, repeated thresholds, the map update
    riding in every rank's weight kernels, two steps in a row): scores, pose blocks, weights, slots, best
    index, counters, poses and the grid of every rank against the one-GPU handle, bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_shard.py"), "20", "4"],
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-300:]
