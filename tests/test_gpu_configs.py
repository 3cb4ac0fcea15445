"""GPU tests (-m gpu) at BASELINE.json's full per-GPU sizes (configs[2], [3], [4]; configs[0] and [1]
are in test_gpu_parity.py): the HIP path through the C ABI against the oracle on a sample of
particles (scores, maps) and on ALL resampled slots, plus size-independent properties over every
particle (integer scores are additive over a split of the beam list; every replaced slot's ancestor
survives; weights sum to 1; log-weights = log of the weights within 1e-5 relative). Bar: bit-exact for
scores, voxel indices, maps and slots. One GPU's share of the 8-GPU configs (the driver runs the
8-GPU jobs): 125 000 x 16 384 on 256^3, 12 500 x 131 072 on 1024^3; configs[3] whole: 50 000
per-particle 128^3 maps = 97.7 GiB of HBM."""
import numpy as np
import pytest

from oracle import binding as O
from slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _wall_beams(grid, G, M, true_vox, yaw0, rng):
    """Local voxels of randomly chosen wall cells seen from the true pose (no ray trace: a 1024^3
    trace in numpy takes minutes), distinct, in the order of S3P:868."""
    occ = np.argwhere(grid[::2, ::2, ::2] == 100) * 2 if G > 256 else np.argwhere(grid == 100)
    pick = occ[rng.choice(len(occ), min(len(occ), 3 * M), replace=False)].astype(np.float64)
    d = pick - true_vox
    c, s = np.cos(-yaw0), np.sin(-yaw0)
    loc = np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1], d[:, 2]], 1)
    loc = np.unique(np.rint(loc).astype(np.int32), axis=0)[:M]
    return tuple(np.ascontiguousarray(loc[:, k]) for k in range(3)) + (np.full(len(loc), 50, np.int8),)


def _shared_share(P, M, G):
    dims = (G, G, G)
    rng = np.random.default_rng(11)
    grid = synth.make_room(dims, seed=1)
    res = 0.05
    true_vox = np.array([G / 2 + 0.3, G / 2 - 0.2, G / 2 + 0.1]); yaw0 = 0.35
    bx, by, bz, bv = _wall_beams(grid, G, M, true_vox, yaw0, rng)
    M = len(bx)
    x = true_vox[0] * res + rng.normal(0, 0.03, P); y = true_vox[1] * res + rng.normal(0, 0.03, P)
    z = true_vox[2] * res + rng.normal(0, 0.015, P); yaw = yaw0 + rng.normal(0, 0.02, P)
    h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=G, grid_y=G, grid_z=G,
                                        local_x=0, local_y=0, local_z=0, max_beams=M))
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_weights(np.full(P, 1.0 / P))
        # additivity over a split of the beam list, for every particle (integer sums: exact)
        half = M // 2
        h.set_beams(bx[:half], by[:half], bz[:half], bv[:half]); h.measure(0); s1, c1 = h.get_scores()
        h.set_beams(bx[half:], by[half:], bz[half:], bv[half:]); h.measure(0); s2, c2 = h.get_scores()
        h.set_beams(bx, by, bz, bv); h.measure(0); sg, cg = h.get_scores()
        assert np.array_equal(s1 + s2, sg) and np.array_equal(c1 + c2, cg)
        assert cg.max() <= M and sg.min() >= 0 and sg.max() <= 100 * M
        # a sample of particles against the oracle: scores and the voxel index of every beam
        pick = np.arange(0, P, max(1, P // 32))
        so, co = O.measure(O.ref_grid(*dims), x[pick], y[pick], z[pick], yaw[pick], grid.reshape(-1), 0,
                           bx, by, bz, bv, 0)
        assert np.array_equal(sg[pick], so) and np.array_equal(cg[pick], co)
        for i in pick[:3]:
            assert np.array_equal(h.voxel_indices(int(i)),
                                  O.voxel_indices(O.ref_grid(*dims), x[i], y[i], z[i], yaw[i], bx, by, bz))
        # weights and ALL resampled slots against the oracle
        u = capi.draw_uniform(4, P)
        h.weights()
        w_pre, logw = h.get_weights()
        wo = np.full(P, 1.0 / P); O.weights(wo, sg, int(cg[-1]))
        assert abs(w_pre.sum() - 1.0) < 1e-12 and np.allclose(w_pre, wo, rtol=1e-12, atol=0)
        ok = w_pre > 0
        assert np.allclose(logw[ok], np.log(w_pre[ok]), rtol=1e-5, atol=1e-6)
        slots, best, nohit, amb = h.resample(u)
        _, sa, ob, onh = O.resample(w_pre.copy(), u, 0)
        assert np.array_equal(slots, sa) and best == ob and nohit == onh == 0
        # properties: survivors keep their slot, every ancestor is a survivor
        assert np.array_equal(slots[slots], slots)
        px, py, pz, pyaw = h.get_poses()
        assert np.array_equal(px, x[slots]) and np.array_equal(pyaw, yaw[slots])
        w_post, _ = h.get_weights()
        assert np.array_equal(w_post, w_pre[slots])
    finally:
        h.close()


def test_config2_share_1m_particles_over_8_gpus():
    """BASELINE configs[2]: 1 M particles x 16 k-point scans over 8 GPUs -> 125 000 x 16 384 per GPU,
    256^3 shared grid replica."""
    _shared_share(125000, 16384, 256)


def test_config4_share_1024_grid_130k_beams():
    """BASELINE configs[4]: 1024^3 grid (1 GiB + its gather-layout copy), ~130 k-point sweeps, 100 k
    particles over 8 GPUs -> 12 500 x 131 072 per GPU."""
    _shared_share(12500, 131072, 1024)


def test_config3_per_particle_maps_50k():
    """BASELINE configs[3]: FastSLAM-style per-particle 128^3 maps, 50 000 particles (97.7 GiB):
    measure + fuse with hits and misses, weights, resample moving the 2 MiB slabs."""
    P, M = 50000, 2048
    dims = (128, 128, 128)
    sc = synth.scene(dims, P, M, seed=5)
    bx, by, bz, bv = sc["beams"]
    rng = np.random.default_rng(1)
    bv = rng.choice(np.array([-30, -3, 10, 30, 50], np.int8), len(bx))
    h = capi.Handle(capi.default_config(n_particles=P, grid_x=128, grid_y=128, grid_z=128,
                                        local_x=0, local_y=0, local_z=0, max_beams=len(bx)))
    try:
        h.broadcast_map(sc["grid"]); h.predict(sc["x"], sc["y"], sc["z"], sc["yaw"])
        h.set_weights(np.full(P, 1.0 / P)); h.set_beams(bx, by, bz, bv)
        h.measure(1)
        s, c = h.get_scores()
        pick = np.arange(0, P, max(1, P // 40))
        maps = np.tile(sc["grid"].reshape(1, -1), (len(pick), 1))
        so, co = O.measure(O.ref_grid(*dims), sc["x"][pick], sc["y"][pick], sc["z"][pick], sc["yaw"][pick],
                           maps.reshape(-1), maps.shape[1], bx, by, bz, bv, 1)
        assert np.array_equal(s[pick], so) and np.array_equal(c[pick], co)
        for k, i in enumerate(pick[:10]):
            assert np.array_equal(h.read_map(int(i)), maps[k]), i
        h.weights()
        w_pre, _ = h.get_weights()
        wo = np.full(P, 1.0 / P); O.weights(wo, s, int(c[-1]))
        assert np.allclose(w_pre, wo, rtol=1e-12, atol=0)
        u = capi.draw_uniform(4, P)
        slots, best, nohit, amb = h.resample(u)
        _, sa, ob, onh = O.resample(w_pre.copy(), u, 0)
        assert np.array_equal(slots, sa) and best == ob and nohit == onh == 0
        moved = np.flatnonzero(slots != np.arange(P))
        assert len(moved) > P // 10
        # a moved slab equals its ancestor's, and (for sampled ancestors) the oracle's fused map
        for i in moved[:: max(1, len(moved) // 6)][:6]:
            assert np.array_equal(h.read_map(int(i)), h.read_map(int(slots[i])))
        where = {int(p): k for k, p in enumerate(pick)}
        hit = [int(i) for i in moved if int(slots[i]) in where][:4]
        for i in hit:
            assert np.array_equal(h.read_map(i), maps[where[int(slots[i])]])
    finally:
        h.close()


def test_reference_resample_cases_on_the_gpu(golden_dir):
    """The 20 resample() cases of the reference itself (tests/golden/resample.npz: flat, skewed, zeros,
    one-hot-like, unnormalised weights; P = 50 and 100) through pf_set_weights + pf_resample, with the
    reference's previous index_high_weight (S3P:756) restored first."""
    import os
    g = np.load(os.path.join(golden_dir, "resample.npz"))
    names = sorted(k[:-2] for k in g.files if k.endswith("_u"))
    assert len(names) >= 16
    handles = {}
    try:
        for n in names:
            w = g[n + "_w"]
            P = len(w)
            if P not in handles:
                handles[P] = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=8,
                                                             grid_y=8, grid_z=8, local_x=0, local_y=0, local_z=0))
            h = handles[P]
            tag = np.arange(P, dtype=np.float64)
            h.predict(np.zeros(P), np.zeros(P), tag, np.zeros(P))       # the fixture tags particles by pos_z
            h.set_weights(w)
            h.set_best_index(int(g[n + "_prev_best"]))
            slots, best, nohit, amb = h.resample(g[n + "_u"])
            assert nohit == 0 and amb == 0, n
            assert np.array_equal(slots, g[n + "_slot_anc"]), n
            assert best == int(g[n + "_best"]), n
            wa, _ = h.get_weights()
            assert np.array_equal(wa, g[n + "_w_after"]), n
            assert np.array_equal(h.get_poses()[2], g[n + "_slot_anc"].astype(np.float64)), n
    finally:
        for h in handles.values():
            h.close()


@pytest.mark.parametrize("P", [64, 5000])
def test_previous_best_survives_a_step_without_positive_weights(P):
    """S3P:815-820: with no weight above 0 (all zero, or NaN from num_match == 0, S3P:928) the loop
    never assigns index_high_weight: the previous NON-ZERO index is kept, as the oracle keeps it."""
    h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=8, grid_y=8, grid_z=8,
                                        local_x=0, local_y=0, local_z=0))
    try:
        u = capi.draw_uniform(9, P)
        prev = P - 7
        for w in (np.zeros(P), np.full(P, np.nan)):
            h.set_best_index(prev)
            h.set_weights(w)
            slots, best, nohit, amb = h.resample(u)
            wo = w.copy()
            _, sa, ob, onh = O.resample(wo, u, prev)
            assert best == ob == prev and nohit == onh == P
            assert np.array_equal(slots, sa) and np.array_equal(slots, np.arange(P))
        # ... and a later step with positive weights replaces it
        w = np.zeros(P); w[3] = 0.25; w[P // 2] = 0.75
        h.set_weights(w)
        slots, best, nohit, amb = h.resample(u)
        _, sa, ob, onh = O.resample(w.copy(), u, prev)
        assert np.array_equal(slots, sa) and best == ob and best != prev
    finally:
        h.close()


@pytest.mark.parametrize("P,spread", [(3000, 1.0), (2500, 12.0), (5000, 0.2)])
def test_lds_box_measure_kernel_matches_oracle(P, spread, monkeypatch):
    """k_measure_box (SLAM_PF_K1=box: gathers from LDS-staged boxes of the grid): scores and voxel
    indices equal the oracle's on a tight cloud, on a cloud spread so wide that boxes overflow LDS,
    and with particles pushed to the grid's edge (lanes off an axis: the reference's flat-only
    bounds test, S3P:882)."""
    monkeypatch.setenv("SLAM_PF_K1", "box")
    dims = (128, 128, 64)
    sc = synth.scene(dims, P, 1500, seed=7)
    rng = np.random.default_rng(P)
    tp = sc["true_pose"]
    x = tp[0] + rng.normal(0, 0.03 * spread, P); y = tp[1] + rng.normal(0, 0.03 * spread, P)
    z = tp[2] + rng.normal(0, 0.015 * spread, P); yaw = tp[3] + rng.normal(0, 0.02 * spread, P)
    x[:40] = rng.uniform(-0.3, 0.4, 40); y[40:80] = dims[1] * 0.05 + rng.uniform(-0.4, 0.3, 40)   # at the edges
    z[80:100] = rng.uniform(-0.5, 0.2, 20)
    z[100] = 1.475; z[101] = 0.525                                                             # z/res = k + 0.5: ambiguous shift
    bx, by, bz, bv = sc["beams"]
    h = capi.Handle(capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                                        grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=len(bx)))
    try:
        h.write_map(0, sc["grid"]); h.predict(x, y, z, yaw); h.set_beams(bx, by, bz, bv)
        h.measure(0)
        s, c = h.get_scores()
        so, co = O.measure(O.ref_grid(*dims), x, y, z, yaw, sc["grid"].reshape(-1), 0, bx, by, bz, bv, 0)
        assert np.array_equal(s, so) and np.array_equal(c, co)
    finally:
        h.close()
