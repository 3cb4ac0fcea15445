"""TEST INFRASTRUCTURE — ctypes view of oracle/libpf_oracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (slam_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libpf_oracle.so")


class Grid(C.Structure):
    _fields_ = [("gx", C.c_int32), ("gy", C.c_int32), ("gz", C.c_int32),
                ("resolution", C.c_double), ("occ_threshold", C.c_int32),
                ("map_lo", C.c_int32), ("map_hi", C.c_int32)]


class UpdateCfg(C.Structure):
    _fields_ = [("gx", C.c_int32), ("gy", C.c_int32), ("gz", C.c_int32),
                ("resolution", C.c_double), ("is_signed", C.c_int32),
                ("hit_inc", C.c_int32), ("hit_clamp", C.c_int32),
                ("miss_inc", C.c_int32), ("miss_clamp", C.c_int32),
                ("max_range", C.c_double)]


def build(force=False):
    src = os.path.join(_HERE, "pf_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-std=c11", "-shared",
                               "-o", _LIB, src, "-lm"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.pfo_snapshot_beams.restype = C.c_int64
        _lib.pfo_weights.restype = C.c_double
        _lib.pfo_resample.restype = C.c_int64
        _lib.pfo_scan_vote.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def ref_grid(gx=300, gy=300, gz=80, resolution=0.05, occ_threshold=20, map_lo=0, map_hi=100):
    return Grid(gx, gy, gz, resolution, occ_threshold, map_lo, map_hi)


def snapshot_beams(local, origin=(100, 100, 30)):
    """local: int8 array [lgx, lgy, lgz]. Returns (bx, by, bz, bv)."""
    local = _c(local, np.int8)
    lgx, lgy, lgz = local.shape
    cap = int(np.count_nonzero(local))
    bx = np.empty(cap, np.int32); by = np.empty(cap, np.int32); bz = np.empty(cap, np.int32)
    bv = np.empty(cap, np.int8)
    n = lib().pfo_snapshot_beams(_p(local), lgx, lgy, lgz, origin[0], origin[1], origin[2],
                                 _p(bx), _p(by), _p(bz), _p(bv), C.c_int64(cap))
    assert n == cap
    return bx, by, bz, bv


def measure(grid, x, y, z, yaw, maps, map_stride, bx, by, bz, bv, add_points):
    """maps: uint8 array modified in place when add_points. Returns (sum, cnt)."""
    P = len(x)
    assert maps.dtype == np.uint8 and maps.flags.c_contiguous
    x = _c(x, np.float64); y = _c(y, np.float64); z = _c(z, np.float64); yaw = _c(yaw, np.float64)
    bx = _c(bx, np.int32); by = _c(by, np.int32); bz = _c(bz, np.int32); bv = _c(bv, np.int8)
    s = np.zeros(P, np.int32); c = np.zeros(P, np.int32)
    lib().pfo_measure(C.byref(grid), C.c_int64(P), _p(x), _p(y), _p(z), _p(yaw),
                      _p(maps), C.c_size_t(map_stride), C.c_int64(len(bx)),
                      _p(bx), _p(by), _p(bz), _p(bv), int(add_points), _p(s), _p(c))
    return s, c


def voxel_indices(grid, x, y, z, yaw, bx, by, bz):
    bx = _c(bx, np.int32); by = _c(by, np.int32); bz = _c(bz, np.int32)
    out = np.empty(len(bx), np.int32)
    lib().pfo_voxel_indices(C.byref(grid), C.c_double(x), C.c_double(y), C.c_double(z),
                            C.c_double(yaw), C.c_int64(len(bx)), _p(bx), _p(by), _p(bz), _p(out))
    return out


def predict(x, y, z, yaw, dl, dyaw, noise):
    """In place on float64 arrays x,y,z,yaw. noise: float32 [P,4]."""
    noise = _c(noise, np.float32)
    for a in (x, y, z, yaw):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    lib().pfo_predict(C.c_int64(len(x)), _p(x), _p(y), _p(z), _p(yaw),
                      C.c_double(dl[0]), C.c_double(dl[1]), C.c_double(dl[2]),
                      C.c_double(dyaw), _p(noise))


def weights(w, s, num_match, want_log=True):
    """In place on float64 w. Returns (total, logw)."""
    assert w.dtype == np.float64 and w.flags.c_contiguous
    s = _c(s, np.int32)
    logw = np.empty(len(w), np.float32) if want_log else None
    tot = lib().pfo_weights(C.c_int64(len(w)), _p(w), _p(s), int(num_match),
                            _p(logw) if want_log else None)
    return tot, logw


def resample(w, u, best):
    """In place on float64 w (permuted). Returns (draw_anc, slot_anc, best, nohit)."""
    assert w.dtype == np.float64 and w.flags.c_contiguous
    u = _c(u, np.float64)
    P = len(w)
    d = np.empty(P, np.int32); sa = np.empty(P, np.int32)
    b = C.c_int32(best)
    nohit = lib().pfo_resample(C.c_int64(P), _p(w), _p(u), _p(d), _p(sa), C.byref(b))
    return d, sa, b.value, nohit


def permute(slot_anc, records):
    """In place on a C-contiguous array whose first axis is the particle axis."""
    assert records.flags.c_contiguous
    slot_anc = _c(slot_anc, np.int32)
    width = records.strides[0]
    lib().pfo_permute(C.c_int64(len(slot_anc)), _p(slot_anc), _p(records), C.c_size_t(width))


def rot_zyx(z, y, x):
    m = np.empty(9, np.float64)
    lib().pfo_rot_zyx(_p(m), C.c_double(z), C.c_double(y), C.c_double(x))
    return m


def update_map(cfg, grid, pos, zyx, points, ray_idx):
    assert grid.flags.c_contiguous and grid.dtype in (np.int8, np.uint8)
    pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
    points = _c(points, np.float64); ray_idx = _c(ray_idx, np.int32)
    lib().pfo_update_map(C.byref(cfg), _p(grid), _p(pos), _p(zyx), C.c_int64(len(points)),
                         _p(points), C.c_int64(len(ray_idx)), _p(ray_idx))


def scan_vote(grid, resolution, points, pos, zyx, n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5,
              length=0.25):
    grid = _c(grid, np.uint8)
    gx, gy, gz = grid.shape
    points = _c(points, np.float64); pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
    sums = np.empty(n_yaw * n_t ** 3, np.int32)
    best = lib().pfo_scan_vote(_p(grid), gx, gy, gz, C.c_double(resolution), C.c_int64(len(points)),
                               _p(points), _p(pos), _p(zyx), n_yaw, yaw_k0, yaw_kstep, n_t,
                               C.c_double(length), _p(sums))
    return sums, best


class Ring(C.Structure):
    _fields_ = [("gx", C.c_int32), ("gy", C.c_int32), ("gz", C.c_int32), ("resolution", C.c_double),
                ("n_slots", C.c_int32), ("cap", C.c_int32),
                ("idx", C.POINTER(C.c_int32)), ("count", C.POINTER(C.c_int32)), ("life", C.POINTER(C.c_int32)),
                ("pos_update", C.c_double * 3), ("euler_update", C.c_double * 3), ("first_run", C.c_int32),
                ("move_threshold", C.c_double), ("recentre_xy", C.c_double), ("recentre_z", C.c_double),
                ("life_new", C.c_int32)]


class RingState:
    """The oracle's keyframe ring (pfo_ring) with numpy-owned storage."""

    def __init__(self, dims, resolution=0.05, n_slots=None, cap=None):
        self.r = Ring()
        lib().pfo_ring_defaults(C.byref(self.r), dims[0], dims[1], dims[2], C.c_double(resolution))
        if n_slots is not None:
            self.r.n_slots = n_slots
        if cap is not None:
            self.r.cap = cap
        self.idx = np.zeros(self.r.n_slots * self.r.cap, np.int32)
        self.count = np.zeros(self.r.n_slots, np.int32)
        self.life = np.zeros(self.r.n_slots, np.int32)
        self.r.idx = self.idx.ctypes.data_as(C.POINTER(C.c_int32))
        self.r.count = self.count.ctypes.data_as(C.POINTER(C.c_int32))
        self.r.life = self.life.ctypes.data_as(C.POINTER(C.c_int32))

    def paint(self, grid):
        assert grid.dtype == np.uint8 and grid.flags.c_contiguous
        lib().pfo_ring_paint(C.byref(self.r), _p(grid))

    def update(self, grid, cloud, pos_scan, zyx):
        assert grid.dtype == np.uint8 and grid.flags.c_contiguous
        cloud = _c(cloud, np.float64).reshape(-1, 3)
        pos = np.array(pos_scan, np.float64); zyx = _c(zyx, np.float64)
        lib().pfo_ring_update.restype = C.c_int
        flags = lib().pfo_ring_update(C.byref(self.r), _p(grid), C.c_int64(len(cloud)), _p(cloud), _p(pos), _p(zyx))
        return flags, pos


def depth_to_cloud(image_bytes, width=640, height=480, pool=4, focal=385.0 / 4.0):
    if isinstance(image_bytes, (bytes, bytearray)):
        image_bytes = np.frombuffer(image_bytes, np.uint8)
    image_bytes = _c(image_bytes, np.uint8).reshape(-1)
    assert len(image_bytes) == width * height * 2
    out = np.empty(((width // pool) * (height // pool), 3), np.float64)
    lib().pfo_depth_to_cloud.restype = C.c_int64
    n = lib().pfo_depth_to_cloud(_p(image_bytes), width, height, pool, C.c_double(focal), _p(out))
    return out[:n].copy()


def read_occupied(grid, resolution=0.05, threshold=80, stride=4, cap=None):
    grid = _c(grid, np.uint8)
    gx, gy, gz = grid.shape
    cap = cap if cap is not None else grid.size // stride + 1
    out = np.empty((cap, 3), np.float32)
    lib().pfo_read_occupied.restype = C.c_int64
    n = lib().pfo_read_occupied(_p(grid), gx, gy, gz, C.c_double(resolution), threshold, stride, _p(out),
                                C.c_int64(cap))
    return out[:n].copy()
