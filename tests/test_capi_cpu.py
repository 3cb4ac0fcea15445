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


def _mt19937_raw(seed, n):
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)                      # init_genrand(seed), what std::mt19937(seed) does
    return bg.random_raw(n).astype(np.uint64)


def test_draw_indices_is_libstdcxx_uniform_int():
    """pfh_draw_indices (S3P:315-319, 666-669): libstdc++ 11's uniform_int_distribution over a
    32-bit engine (bits/uniform_int_dist.h: Lemire's multiply-shift with rejection of the short
    low words), restated here on numpy's raw mt19937 stream; hi is inclusive."""
    assert int(_mt19937_raw(5489, 1)[0]) == 3499211612          # the standard's known first output
    for seed, hi, n in ((7, 19199, 2400), (11, 499, 500), (3, 0, 5), (5, 2, 4000)):
        raw = _mt19937_raw(seed, 2 * n + 64)
        rng_ = hi + 1
        want, k = [], 0
        while len(want) < n:
            prod = int(raw[k]) * rng_; k += 1
            low = prod & 0xFFFFFFFF
            if low < rng_:
                thr = ((1 << 32) - rng_) % rng_
                while low < thr:
                    prod = int(raw[k]) * rng_; k += 1
                    low = prod & 0xFFFFFFFF
            want.append(prod >> 32)
        got = capi.draw_indices(seed, n, hi)
        assert np.array_equal(got, np.array(want, np.int32))
        assert got.min() >= 0 and got.max() <= hi
    assert capi.draw_indices(5, 4000, 2).max() == 2             # inclusive


def test_vote_epilogue_and_pose_message_follow_the_reference_formulas():
    """pfh_vote_epilogue (S3P:379-404) and pfh_pose_message (S3P:207-222) against the formulas
    written out with numpy, all three branches of the gyro unwrap. The epilogue is also pinned to the
    reference build (tests/golden/vote*.npz); the pose message is PARITY UNPINNED: pub_pose_func() fills ROS
    message types, which are not in this image, so the reference build does not contain it."""
    pi = 2 * np.arccos(0.0)
    for gyr, old in ((0.3, 0.1), (-3.0, 3.0), (3.0, -3.0)):
        pos0 = np.array([4.0, 5.0, 1.5]); e0 = 0.7
        loc = np.array([4.1, 4.95, 1.55, 0.7 + 3 * pi / 180])
        pos, yaw, new_old, pd, ds, d = capi.vote_epilogue(pos0, e0, loc, gyr, old)
        dx, dy = loc[0] - pos0[0], loc[1] - pos0[1]
        want_pd = np.array([np.cos(-e0) * dx - np.sin(-e0) * dy, np.sin(-e0) * dx + np.cos(-e0) * dy,
                            loc[2] - pos0[2]])
        assert np.array_equal(pd, want_pd)
        assert np.array_equal(pos, loc[:3]) and yaw == loc[3] and new_old == gyr
        if abs(old - gyr) < pi:
            dg = gyr - old
        elif gyr < 0:
            dg = 2 * pi + gyr - old
        else:
            dg = -2 * pi + gyr - old
        assert ds == loc[3] - e0 and d == (dg + ds) / 2
    p, q = capi.pose_message([1.0, 2.0, 3.0], 0.4)
    assert np.array_equal(p, [1.0, -3.0, 2.0])
    a = (-0.4 - pi / 2) / 2
    assert np.array_equal(q, [np.cos(a), 0.0, np.sin(a), 0.0])


def test_host_trig_is_the_merged_sincos_of_the_gxx_build():
    """g++ -O2 merges the reference's sin(a) and cos(a) into glibc's sincos(a), which differs from
    sin()/cos() in the last place for about one argument in a thousand (slam_amd/csrc/pf_trig.h); the
    library and the oracle both take the pair from sincos(). Checked on arguments where the two
    differ: found here with libm itself, 0.40005245828215963 among them (the yaw at which a whole-step
    fuzz case first showed it, through a ray length on a rounding tie)."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.cos.restype = libm.sin.restype = C.c_double
    libm.cos.argtypes = libm.sin.argtypes = [C.c_double]
    libm.sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rng = np.random.default_rng(3)
    cand = np.concatenate([[0.40005245828215963], rng.uniform(-3.2, 3.2, 40000)])
    yaws = []
    for a in cand:
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(float(a), C.byref(sv), C.byref(cv))
        if sv.value != libm.sin(float(a)) or cv.value != libm.cos(float(a)):
            yaws.append(float(a))
    if len(yaws) < 5:
        pytest.skip("this libm's sincos() agrees with sin() and cos(): nothing to tell apart")
    yaw = np.array(yaws); P = len(yaw)
    st = [np.zeros(P), np.zeros(P), np.full(P, 1.5), yaw]      # from the origin: the last place stays visible
    s1 = [v.copy() for v in st]; s2 = [v.copy() for v in st]
    noise = np.zeros((P, 4), np.float32)
    capi.host_predict(*s1, (0.12, -0.07, 0.02), 0.0, noise)
    O.predict(*s2, (0.12, -0.07, 0.02), 0.0, noise)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    plain = np.cos(yaw) * 0.12 - np.sin(yaw) * -0.07
    assert np.any(plain != s1[0])        # the separate calls would have been visible here
    for a in yaws[:50]:                  # pfh_update_realtime_pos: one sincos for S3P:713-714
        sv, cv = C.c_double(), C.c_double()
        libm.sincos(a, C.byref(sv), C.byref(cv))
        pos, _ = capi.update_realtime_pos(np.zeros(3), a, np.array([0.3, -0.2, 0.0]), 0.0, np.zeros(4))
        assert pos[0] == cv.value * 0.3 - sv.value * -0.2
        assert pos[1] == sv.value * 0.3 + cv.value * -0.2


@pytest.mark.parametrize("world,P", [(2, 50), (4, 33), (8, 125)])
def test_migration_plan_moves_every_remote_ancestor(world, P):
    """pfh_migration_plan (S3P:805-812 across shards): each rank's send and receive lists, played
    through per-pair FIFO channels the way RCCL matches sends and receives of one pair of ranks,
    must leave every slot with its ancestor's slab; transfer numbers agree on both ends."""
    rng = np.random.default_rng(world * 1000 + P)
    Pg = world * P
    w = rng.random(Pg) ** 6 + 1e-9
    w /= w.sum()
    _, slot_anc, _, _ = O.resample(w, rng.random(Pg), 0)
    slot_anc = np.asarray(slot_anc, np.int32)
    assert np.array_equal(slot_anc[slot_anc], slot_anc)          # sources are surviving slots
    slabs = [np.arange(r * P, (r + 1) * P) for r in range(world)]   # slab tag = its first owner
    plans = [capi.migration_plan(slot_anc, P, r) for r in range(world)]
    totals = {p[0] for p in plans}
    assert len(totals) == 1
    remote = sum(1 for g in range(Pg) if slot_anc[g] != g and slot_anc[g] // P != g // P)
    assert totals.pop() == remote and remote > 0
    new = [s.copy() for s in slabs]
    for r in range(world):                                       # local copies (k_apply_maps)
        for i in range(P):
            a = int(slot_anc[r * P + i])
            if a // P == r:
                new[r][i] = slabs[r][a - r * P]
    chan = {}
    for r, (_, sends, _) in enumerate(plans):
        assert [q for _, _, q in sends] == sorted(q for _, _, q in sends)
        for src, peer, seq in sends:
            assert peer != r
            chan.setdefault((r, peer), []).append((slabs[r][src], seq))
    for r, (_, _, recvs) in enumerate(plans):
        assert [q for _, _, q in recvs] == sorted(q for _, _, q in recvs)
        for dst, peer, seq in recvs:
            tag, sseq = chan[(peer, r)].pop(0)
            assert sseq == seq
            new[r][dst] = tag
    assert all(len(v) == 0 for v in chan.values())
    assert np.array_equal(np.concatenate(new), slot_anc)


def test_deal_groups_gives_every_group_to_exactly_one_chunk():
    """pfh_deal_groups, the host logic behind K1's chunk-major group list (pf_set_beams): the list it lays out is a
    permutation of the sorted one, every chunk's share is contiguous and ordered by count, the table's ends fit, the trips
    of a count are dealt round-robin through the counts (shares differ by at most one trip per count and over the whole
    list), and a short last trip stays in one chunk."""
    rng = np.random.default_rng(4)
    cases = [([0] * 8, 1, [1] * 8), ([5] * 8, 3, [2] * 8), ([0, 0, 0, 0, 0, 0, 0, 7], 51, [1, 1, 1, 1, 2, 2, 2, 2])]
    for _ in range(60):
        counts = rng.integers(0, 200, 8) * (rng.random(8) < 0.8)
        cases.append((np.cumsum(counts).tolist(), int(rng.integers(1, 70)), rng.integers(1, 5, 8).tolist()))
    for ends, C_, trips in cases:
        perm, table = capi.deal_groups(ends, C_, trips)
        n = ends[7]
        assert sorted(perm.tolist()) == list(range(n))
        count_of = np.zeros(n, np.int64)                      # phase (0: eight beams ... 7: one) of every sorted group
        g0 = 0
        for ph in range(8):
            count_of[g0:ends[ph]] = ph; g0 = ends[ph]
        at = 0
        trips_of = np.zeros((C_, 8), np.int64)
        for y in range(C_):
            assert table[y, 8] == at
            for ph in range(8):
                seg = perm[at:table[y, ph]]
                assert np.all(count_of[seg] == ph)
                assert np.all(np.diff(seg) > 0)               # list order kept inside a chunk's count
                T = trips[ph]
                first = ends[ph - 1] if ph else 0
                # whole trips: a chunk's groups of a count come in runs of T consecutive sorted groups from a trip boundary
                for k in range(0, len(seg), T):
                    run = seg[k:k + T]
                    assert (run[0] - first) % T == 0 and np.all(np.diff(run) == 1)
                    assert len(run) == T or run[-1] == ends[ph] - 1
                trips_of[y, ph] = (len(seg) + T - 1) // T
                at = int(table[y, ph])
        assert at == n
        assert np.all(trips_of.max(0) - trips_of.min(0) <= 1)
        assert trips_of.sum(1).max() - trips_of.sum(1).min() <= 1
