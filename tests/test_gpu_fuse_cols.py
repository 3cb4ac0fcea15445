"""GPU suite: the column form of K1+K2 on per-particle maps (slam_amd/csrc/pf_fuse_cols.hip, S3P:859-908
with add_points on) against the oracle, on the cases its design singles out: columns taller than one
entry can hold, holes inside a run, particles whose columns leave the map (the flat-only bounds test of
S3P:882 then aliases cells, the particle is walked beam by beam), z shifts on a rounding tie, thresholds
below zero, map cells outside [map_lo, map_hi] that no beam touches, beam lists that are not in the
local map's x, y, z order and hit a cell twice. Every score and every map byte is compared; the
kernel's own statistics (SLAM_PF_FCOL_STATS) show which route the particles took."""
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


def _dense_columns(rng, half, z_lo, z_hi, hole=0.1):
    bx, by, bz, bv = [], [], [], []
    for lx in range(-half, half + 1):
        for ly in range(-half, half + 1):
            lo = z_lo + int(rng.integers(0, 6)); hi = z_hi - int(rng.integers(0, 6))
            for lz in range(lo, hi + 1):
                if rng.random() < hole:
                    continue
                bx.append(lx); by.append(ly); bz.append(lz)
                v = 0
                while v == 0:
                    v = int(rng.integers(-60, 61))
                bv.append(v)
    return (np.array(bx, np.int32), np.array(by, np.int32), np.array(bz, np.int32), np.array(bv, np.int8))


def _run(capfd, monkeypatch, dims, beams, x, y, z, yaw, grid0, steps=2, **cfg_kw):
    from slam_amd import capi
    from oracle import binding as O
    monkeypatch.setenv("SLAM_PF_FCOL_STATS", "1")
    monkeypatch.setenv("SLAM_PF_FUSE_COLS", "1")           # also for lists too sparse for the automatic choice
    P = len(x)
    og = O.ref_grid(*dims, occ_threshold=cfg_kw.get("occ_threshold", 20), map_lo=cfg_kw.get("map_lo", 0),
                    map_hi=cfg_kw.get("map_hi", 100))
    maps = np.tile(grid0.reshape(1, -1), (P, 1))
    cfg = capi.default_config(n_particles=P, grid_x=dims[0], grid_y=dims[1], grid_z=dims[2], local_x=0, local_y=0,
                              local_z=0, max_beams=len(beams[0]), **cfg_kw)
    h = capi.Handle(cfg)
    routes = []
    try:
        h.broadcast_map(grid0); h.predict(x, y, z, yaw); h.set_beams(*beams)
        for step in range(steps):
            s, c = O.measure(og, x, y, z, yaw, maps.reshape(-1), maps.shape[1], *beams, 1)
            capfd.readouterr()
            h.measure(1)
            gs, gc = h.get_scores()
            err = capfd.readouterr().err
            m = re.search(r"\[fuse_cols\] P=(\d+) columns=(\d+): (\d+) particles beam by beam", err)
            assert m, "the column kernel did not run: " + err[-300:]
            routes.append((int(m.group(2)), int(m.group(3))))
            assert np.array_equal(gs, s) and np.array_equal(gc, c), step
            for i in range(P):
                assert np.array_equal(h.read_map(i), maps[i]), (step, i)
        # measure alone (S3P:892 add_points == 0) on the fused maps: the occupied beams' column form
        s, c = O.measure(og, x, y, z, yaw, maps.reshape(-1), maps.shape[1], *beams, 0)
        capfd.readouterr()
        h.measure(0)
        gs, gc = h.get_scores()
        assert "[fuse_cols]" in capfd.readouterr().err
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        assert np.array_equal(h.read_map(0), maps[0])
    finally:
        h.close()
    return routes


def test_tall_columns_holes_edges_and_ties(capfd, monkeypatch):
    """13 x 13 local columns up to 86 cells tall (more than PF_FCOL_MAXZ: two entries per column) with holes;
    most particles in the middle of a 48 x 48 x 96 map at every yaw, some at the rim (columns leave the map
    on x or y), some with their z shift exactly on a tie, some too high (columns leave on z)."""
    rng = np.random.default_rng(5)
    dims = (48, 48, 96)
    beams = _dense_columns(rng, 6, -40, 45)
    P = 96
    res = 0.05
    x = rng.uniform(1.1, 1.3, P); y = rng.uniform(1.1, 1.3, P); z = rng.uniform(2.25, 2.45, P)
    yaw = rng.uniform(-3.2, 3.2, P)
    x[:8] = rng.uniform(0.1, 0.3, 8)                       # rim: some columns have X < 0
    y[8:12] = 2.3                                          # rim on y
    z[12:18] = (np.arange(44, 50) + 0.5) * res             # z / res on a rounding tie
    z[18:22] = 3.0                                         # 60 + 45 > 95: off the top
    grid0 = rng.integers(0, 256, dims).astype(np.uint8)    # cells beyond map_hi too
    routes = _run(capfd, monkeypatch, dims, beams, x, y, z, yaw, grid0, occ_threshold=-3, map_lo=5, map_hi=100)
    ncols, slow = routes[0]
    assert ncols > 13 * 13                                 # tall columns were cut
    assert 12 <= slow < P                                  # rim, tie and top particles beam by beam, the rest by columns


def test_lists_out_of_order_with_repeated_cells(capfd, monkeypatch):
    """A beam list in random order in which a tenth of the cells appear twice: the columns are single
    beams or short runs, many share (lx, ly), and the order of the list decides the clamped sums."""
    rng = np.random.default_rng(6)
    dims = (40, 40, 32)
    bx, by, bz, bv = _dense_columns(rng, 5, -12, 12, hole=0.3)
    extra = rng.choice(len(bx), len(bx) // 10, replace=False)
    bx = np.concatenate([bx, bx[extra]]); by = np.concatenate([by, by[extra]]); bz = np.concatenate([bz, bz[extra]])
    bv = np.concatenate([bv, rng.choice(np.array([-70, -20, 15, 40, 90], np.int8), len(extra))])
    perm = rng.permutation(len(bx))
    beams = (bx[perm], by[perm], bz[perm], bv[perm])
    P = 40
    x = rng.uniform(0.9, 1.1, P); y = rng.uniform(0.9, 1.1, P); z = rng.uniform(0.7, 0.9, P)
    yaw = rng.uniform(-3.2, 3.2, P)
    grid0 = rng.integers(30, 71, dims).astype(np.uint8)
    routes = _run(capfd, monkeypatch, dims, beams, x, y, z, yaw, grid0, steps=3)
    assert routes[0][1] == 0                               # every particle by columns, the election keeps the order


def test_saturated_maps_stay_put(capfd, monkeypatch):
    """Cells at their clamp are not rewritten (S3P:896-901 leaves them where they are): a second and third
    fuse of the same beams over maps that saturate must still match byte for byte."""
    rng = np.random.default_rng(7)
    dims = (32, 32, 32)
    beams = _dense_columns(rng, 4, -10, 10, hole=0.05)
    P = 24
    x = rng.uniform(0.75, 0.85, P); y = rng.uniform(0.75, 0.85, P); z = rng.uniform(0.75, 0.85, P)
    yaw = rng.uniform(-0.4, 0.4, P)
    grid0 = np.full(dims, 50, np.uint8)
    _run(capfd, monkeypatch, dims, beams, x, y, z, yaw, grid0, steps=4)


def test_k1_run_groups_match_oracle():
    """K1 on the shared grid (k_measure_col) with columns dense enough for its RUN groups (8-cell z windows
    read as three dwords, pf_api.hip pf_set_beams): walls as tall z runs with holes, single cells, a cell
    listed twice (no run for that column), particles in the middle at every yaw, at the rim, too high, too
    low and on a z tie. Every score against the oracle."""
    from slam_amd import capi, synth
    from oracle import binding as O
    rng = np.random.default_rng(11)
    dims = (64, 64, 48)
    bx, by, bz, bv = _dense_columns(rng, 7, -14, 14, hole=0.15)
    bv = np.where(rng.random(len(bv)) < 0.7, np.int8(60), np.int8(5)).astype(np.int8)      # 30 % not occupied
    bx = np.concatenate([bx, bx[:3]]); by = np.concatenate([by, by[:3]]); bz = np.concatenate([bz, bz[:3]])
    bv = np.concatenate([bv, np.full(3, 60, np.int8)])                                         # repeated cells
    grid = synth.make_room(dims, 1)
    grid[rng.random(dims) < 0.3] = 77
    P = 700
    x = rng.uniform(1.5, 1.7, P); y = rng.uniform(1.5, 1.7, P); z = rng.uniform(1.1, 1.3, P)
    yaw = rng.uniform(-3.2, 3.2, P)
    x[:40] = rng.uniform(0.1, 0.4, 40); y[40:60] = 3.0; z[60:90] = 2.0; z[90:120] = 0.3
    z[120:140] = (np.arange(20, 40) + 0.5) * 0.05
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                              grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=len(bx))
    h = capi.Handle(cfg)
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_beams(bx, by, bz, bv)
        h.measure(0)
        gs, gc = h.get_scores()
        g1 = grid.reshape(-1).copy()
        s, c = O.measure(O.ref_grid(*dims), x, y, z, yaw, g1, 0, bx, by, bz, bv, 0)
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
        assert len(np.unique(gc)) > 3
    finally:
        h.close()


def test_k1_phase_that_is_empty_after_an_odd_boundary():
    """k_measure_col deals pairs of column groups to its chunks across the phase boundaries (4-, 3-, 2-,
    1-beam groups). With an odd number of 4-beam groups and no 3-beam group, the pair slot on the boundary
    holds nothing of the 3-beam phase: it used to run a trip there and counted the next group twice (found
    by tools/fuzz_fuse_cols.py). Three columns of four beams spread over z (no run group), five single
    beams, against the oracle."""
    from slam_amd import capi, synth
    from oracle import binding as O
    dims = (48, 48, 48)
    bx, by, bz = [], [], []
    for lx in (-3, 0, 4):
        for lz in (-12, -2, 7, 15):
            bx.append(lx); by.append(2); bz.append(lz)
    for ly in range(-4, 1):
        bx.append(6); by.append(ly); bz.append(1)
    bx, by, bz = (np.array(a, np.int32) for a in (bx, by, bz))
    order = np.lexsort((bz, by, bx))
    bx, by, bz = bx[order], by[order], bz[order]
    bv = np.full(len(bx), 50, np.int8)
    rng = np.random.default_rng(3)
    grid = rng.integers(0, 101, dims).astype(np.uint8)
    P = 300
    x = rng.uniform(1.1, 1.3, P); y = rng.uniform(1.1, 1.3, P); z = rng.uniform(1.1, 1.3, P); yaw = rng.uniform(-3.2, 3.2, P)
    cfg = capi.default_config(n_particles=P, map_mode=capi.PF_MAP_SHARED, grid_x=dims[0], grid_y=dims[1],
                              grid_z=dims[2], local_x=0, local_y=0, local_z=0, max_beams=len(bx))
    h = capi.Handle(cfg)
    try:
        h.write_map(0, grid); h.predict(x, y, z, yaw); h.set_beams(bx, by, bz, bv)
        h.measure(0)
        gs, gc = h.get_scores()
        g1 = grid.reshape(-1).copy()
        s, c = O.measure(O.ref_grid(*dims), x, y, z, yaw, g1, 0, bx, by, bz, bv, 0)
        assert np.array_equal(gs, s) and np.array_equal(gc, c)
    finally:
        h.close()


def test_random_scenes_through_the_column_kernels():
    """tools/fuzz_fuse_cols.py, a few dozen random scenes (grid heights 16 to 112, dense/holed/sparse/wall
    lists, repeats and disorder, particles at the rim and on z ties, thresholds and clamps varied): the
    column kernels with and without the add on per-particle maps and K1's run groups on the shared grid,
    every score and map byte against the oracle. In a process of its own: it forces the column form."""
    import subprocess
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fuse_cols.py"), "40", "11"],
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-300:]


def test_random_whole_steps_on_per_particle_maps():
    """tools/fuzz_pp.py, forty random scenes of three steps on per-particle maps (1 to 400 particles, ragged
    grids, beam lists from a handful to a full box, increments at the int8 limits, disorder and repeats,
    particles on quarter-voxel poses at right-angle yaws): measure with and without the fuse, weights,
    resample with the slabs following their ancestors; every score, weight, slot, pose and map byte against
    the oracle. The library picks its own kernels here (column form or wave per particle)."""
    import subprocess
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_pp.py"), "40", "5"],
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-600:] + out.stderr[-300:]
