"""CPU suite: the C-ABI library loads, exports every symbol include/slam_pf.h declares, refuses
to run without a GPU, and its host-driver helpers (pfh_*) match the oracle."""
import os
import re

import numpy as np
import pytest

from oracle import binding as O
from slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "slam_pf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfh?_[a-z_0-9]+)\s*\(", text)))


def test_header_and_library_agree():
    syms = declared_symbols()
    assert sorted(capi.SYMBOLS) == syms
    L = capi.lib()
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_config_struct_matches_header_defaults():
    cfg = capi.default_config()
    assert cfg.struct_size == __import__("ctypes").sizeof(capi.Config)
    assert (cfg.grid_x, cfg.grid_y, cfg.grid_z) == (300, 300, 80)
    assert (cfg.local_x, cfg.local_y, cfg.local_z) == (200, 200, 80)
    assert cfg.n_particles == 50 and cfg.resolution == 0.05 and cfg.occ_threshold == 20
    assert cfg.map_prior == 50 and (cfg.map_lo, cfg.map_hi) == (0, 100)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PfError, match="no HIP device"):
        capi.Handle(capi.default_config())


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pf_oracle" not in text and "from oracle" not in text and \
                    "import oracle" not in text, f
    assert "oracle" not in open(os.path.join(ROOT, "include", "slam_pf.h")).read()


def test_host_helpers_match_oracle():
    rng = np.random.default_rng(0)
    local = np.zeros((20, 30, 8), np.int8)
    idx = rng.integers(0, local.size, 300)
    local.reshape(-1)[idx] = rng.integers(-50, 51, 300).astype(np.int8)
    a = capi.snapshot_beams(local, (10, 15, 3)); b = O.snapshot_beams(local, (10, 15, 3))
    for p, q in zip(a, b):
        assert np.array_equal(p, q)
    # ascending flat order (S3P:868)
    flat = ((a[0] + 10) * 30 + (a[1] + 15)) * 8 + (a[2] + 3)
    assert np.all(np.diff(flat) > 0)
    empty = capi.snapshot_beams(np.zeros((4, 4, 4), np.int8), (2, 2, 2))
    assert all(len(e) == 0 for e in empty)

    P = 257
    st = [rng.normal(7, 1, P) for _ in range(3)] + [rng.uniform(-3.2, 3.2, P)]
    s1 = [v.copy() for v in st]; s2 = [v.copy() for v in st]
    noise = capi.draw_noise(42, P)
    capi.host_predict(*s1, (0.12, -0.07, 0.02), 0.04, noise)
    O.predict(*s2, (0.12, -0.07, 0.02), 0.04, noise)
    for p, q in zip(s1, s2):
        assert np.array_equal(p, q)

    recs = rng.random((P, 3)); r2 = recs.copy()
    w = rng.random(P); w /= w.sum()
    _, sa, _, _ = O.resample(w.copy(), capi.draw_uniform(5, P), 0)
    capi.permute(sa, recs); O.permute(sa, r2)
    assert np.array_equal(recs, r2) and np.array_equal(recs, rng.random(0).reshape(0, 3)) is False

    last = np.array([7.5, 7.5, 1.5, 0.0])
    assert capi.add_points_policy([7.6, 7.6, 1.5, 0.1], last) == 0 and last[0] == 7.5
    assert capi.add_points_policy([7.8, 7.6, 1.5, 0.1], last) == 1 and last[0] == 7.8


def test_vote_hypothesis_enumeration():
    """pfh_vote_hypothesis walks the lattice in the loop order of S3P:328-348 (r, x, y, z)."""
    pos = np.array([5.01, 4.97, 1.52]); zyx = np.array([0.3, 0.01, -0.02])
    res, length, n_yaw, n_t = 0.05, 0.25, 5, 5
    k = 0
    for r in range(n_yaw):
        rz = zyx[0] + (-6 + 3 * r) * (2 * np.arccos(0.0)) / 180
        for a in range(n_t):
            for b in range(n_t):
                for d in range(n_t):
                    want = [pos[0] + a * res - (length - res) / 2, pos[1] + b * res - (length - res) / 2,
                            pos[2] + d * res - (length - res) / 2, rz]
                    got = capi.vote_hypothesis(k, pos, zyx)
                    assert np.array_equal(got, np.array(want)), k
                    k += 1
    assert capi.vote_hypothesis(-1, pos, zyx) is None and capi.vote_hypothesis(k, pos, zyx) is None


def test_comm_entry_points_fail_loudly_without_a_gpu():
    """pf_comm_unique_id binds RCCL at run time: without a device (or without librccl) it must
    return an error with a message, not crash and not hand out an id."""
    try:
        uid = capi.comm_unique_id()
    except RuntimeError as e:
        assert "pf_comm_unique_id" in str(e)
    else:
        assert len(uid) == 128              # a box with a GPU


def test_resample_oracle_properties():
    """Size-independent properties of the resampler the GPU tests also use."""
    rng = np.random.default_rng(1)
    for P in (1, 2, 7, 64, 1000):
        w = rng.random(P); w[rng.random(P) < 0.3] = 0
        if not w.any():
            w[0] = 1
        w /= w.sum()
        u = rng.random(P)
        w2 = w.copy()
        d, sa, best, nohit = O.resample(w2, u, 0)
        drawn = np.unique(d)
        assert np.array_equal(sa[drawn], drawn)                        # survivors keep their slot
        assert np.array_equal(np.sort(sa), np.sort(d))                 # same multiset of ancestors
        assert np.array_equal(w2, w[sa])
        assert np.all(w[d[np.arange(P)]] > 0) or nohit > 0             # zero-weight never drawn
    # boundary and out-of-range thresholds: no interval matches, the draw keeps its own index
    w = np.array([0.25, 0.25, 0.25, 0.25])
    d, sa, best, nohit = O.resample(w.copy(), np.array([0.0, 0.25, 0.5, 1.0]), 2)
    assert nohit == 4 and np.array_equal(d, np.arange(4)) and np.array_equal(sa, np.arange(4))
