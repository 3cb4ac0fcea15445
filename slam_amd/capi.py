"""ctypes binding of include/slam_pf.h (slam_amd/csrc/libslam_pf.so).

Plumbing only: the product is the shared library. Loading fails loudly when the library has
not been built; there is no Python or CPU substitute for any entry point.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAM_PF_LIB") or os.path.join(_HERE, "csrc", "libslam_pf.so")   # (SLAM_PF_LIB: development builds)

PF_OK = 0
PF_MAP_SHARED, PF_MAP_PER_PARTICLE = 0, 1
PF_ORDER_AUTO, PF_ORDER_EXACT, PF_ORDER_FAST = 0, 1, 2
PF_GRID_SHARED, PF_GRID_LOCAL = 0, 1
PF_K_MEASURE, PF_K_WEIGHTS, PF_K_RESAMPLE, PF_K_UPDATE, PF_K_VOTE, PF_K_ORDER = 0, 1, 2, 3, 4, 5

# every symbol include/slam_pf.h declares (tests check the library exports exactly these)
SYMBOLS = (
    "pf_default_config pf_create pf_destroy pf_last_error pf_predict pf_predict_device pf_get_poses pf_set_weights "
    "pf_get_weights pf_fill_maps pf_write_map pf_broadcast_map pf_read_map pf_set_beams "
    "pf_measure pf_get_scores pf_scores_device pf_voxel_indices pf_weights pf_resample "
    "pf_resample_async pf_resample_result "
    "pf_poses_device pf_set_scan pf_update_map pf_scan_vote pf_write_local_map pf_read_local_map "
    "pf_reset_local_map pf_stream pf_set_stream pf_synchronize pf_profile pf_kernel_time "
    "pfh_snapshot_beams pfh_predict pfh_draw_noise pfh_draw_uniform pfh_add_points_policy "
    "pfh_permute pfh_vote_hypothesis pf_comm_unique_id pf_comm_init "
    "pf_ring_default_params pf_ring_init pf_ring_paint pf_ring_update pf_ring_state "
    "pf_depth_to_cloud pf_read_occupied pf_sequential_cdf pfh_draw_indices pfh_vote_epilogue pfh_pose_message pfh_migration_plan "
    "pfh_update_realtime_pos pf_set_best_index pf_scan_vote_static pf_scan_vote_static_warmup pfh_deal_groups "
    "pf_depth_to_cloud_device pf_scan_vote_device pf_ring_update_device pf_set_scan_device").split()


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32),
                ("n_particles", C.c_int32), ("n_particles_global", C.c_int32),
                ("particle_offset", C.c_int32), ("map_mode", C.c_int32),
                ("grid_x", C.c_int32), ("grid_y", C.c_int32), ("grid_z", C.c_int32),
                ("local_x", C.c_int32), ("local_y", C.c_int32), ("local_z", C.c_int32),
                ("max_beams", C.c_int32), ("max_points", C.c_int32),
                ("occ_threshold", C.c_int32), ("map_lo", C.c_int32), ("map_hi", C.c_int32),
                ("map_prior", C.c_int32), ("order_mode", C.c_int32),
                ("resolution", C.c_double)]


class RingParams(C.Structure):
    _fields_ = [("n_slots", C.c_int32), ("slot_capacity", C.c_int32), ("life_new", C.c_int32),
                ("reserved", C.c_int32), ("move_threshold", C.c_double), ("recentre_xy", C.c_double),
                ("recentre_z", C.c_double)]


class UpdateParams(C.Structure):
    _fields_ = [("hit_inc", C.c_int32), ("hit_clamp", C.c_int32),
                ("miss_inc", C.c_int32), ("miss_clamp", C.c_int32),
                ("max_range", C.c_double)]


class PfError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PfError(f"{LIB_PATH} is missing: build it with `make -C slam_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: torch ships its own libamdhip64 under the same soname as
        # /opt/rocm's, and whichever loads first serves both. With the library's first, torch later
        # finds "No HIP GPUs"; with torch's first, both work. Python callers of this module use torch
        # for device memory and collectives anyway, so bring it in before the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.pf_last_error.restype = C.c_char_p
        L.pf_last_error.argtypes = [C.c_void_p]
        for name in ("pf_scores_device", "pf_poses_device", "pf_stream"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.pfh_snapshot_beams.restype = C.c_int64
        L.pfh_migration_plan.restype = C.c_int64
        L.pf_destroy.restype = None
        L.pf_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    # the address as an integer (a third of the cost of a.ctypes.data_as, and the per-step calls pass ten
    # arrays); every caller here keeps `a` in a local until the call returns
    return None if a is None else C.c_void_p(a.__array_interface__["data"][0])


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def default_config(**kw):
    cfg = Config()
    rc = lib().pf_default_config(C.byref(cfg))
    assert rc == PF_OK
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    if "n_particles" in kw and "n_particles_global" not in kw:
        cfg.n_particles_global = cfg.n_particles
    return cfg


class Handle:
    """One pf_handle: one GPU's shard of the filter. Methods map 1:1 to the C ABI."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = lib().pf_create(C.byref(cfg), C.byref(self._h))
        if rc != PF_OK:
            raise PfError(f"pf_create -> {rc}: {lib().pf_last_error(None).decode()}")
        self.P = cfg.n_particles
        self.Pg = cfg.n_particles_global
        self.elements = cfg.grid_x * cfg.grid_y * cfg.grid_z
        self.local_elements = cfg.local_x * cfg.local_y * cfg.local_z
        self.external_stream = False

    def close(self):
        if self._h:
            lib().pf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != PF_OK:
            raise PfError(f"{what} -> {rc}: {lib().pf_last_error(self._h).decode()}")

    # --- state ---
    def predict(self, x, y, z, yaw):
        x, y, z, yaw = (_c(a, np.float64) for a in (x, y, z, yaw))
        assert len(x) == len(y) == len(z) == len(yaw) == self.P
        self._ck(lib().pf_predict(self._h, _p(x), _p(y), _p(z), _p(yaw)), "pf_predict")

    def comm_init(self, unique_id, rank, world):
        """RCCL communicator for the in-library all-gathers; unique_id: bytes from comm_unique_id()
        of rank 0. Raises when RCCL is not loadable or the init fails."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(lib().pf_comm_init(self._h, buf, int(rank), int(world)), "pf_comm_init")
        self.native_comm = True

    def predict_device(self, pose_block_ptr):
        self._ck(lib().pf_predict_device(self._h, C.c_void_p(pose_block_ptr)), "pf_predict_device")

    def get_poses(self):
        out = [np.empty(self.P, np.float64) for _ in range(4)]
        self._ck(lib().pf_get_poses(self._h, *[_p(a) for a in out]), "pf_get_poses")
        return out

    def set_weights(self, w):
        w = _c(w, np.float64)
        assert len(w) == self.Pg
        self._ck(lib().pf_set_weights(self._h, _p(w)), "pf_set_weights")

    def set_best_index(self, index):
        self._ck(lib().pf_set_best_index(self._h, C.c_int32(index)), "pf_set_best_index")

    def get_weights(self):
        w = np.empty(self.Pg, np.float64)
        lw = np.empty(self.Pg, np.float32)
        self._ck(lib().pf_get_weights(self._h, _p(w), _p(lw)), "pf_get_weights")
        return w, lw

    def fill_maps(self, value):
        self._ck(lib().pf_fill_maps(self._h, int(value)), "pf_fill_maps")

    def write_map(self, i, src):
        src = _c(src, np.uint8).reshape(-1)
        assert src.size == self.elements
        self._ck(lib().pf_write_map(self._h, int(i), _p(src)), "pf_write_map")

    def broadcast_map(self, src):
        src = _c(src, np.uint8).reshape(-1)
        assert src.size == self.elements
        self._ck(lib().pf_broadcast_map(self._h, _p(src)), "pf_broadcast_map")

    def read_map(self, i=0):
        dst = np.empty(self.elements, np.uint8)
        self._ck(lib().pf_read_map(self._h, int(i), _p(dst)), "pf_read_map")
        return dst

    # --- measurement ---
    def set_beams(self, bx, by, bz, bv):
        bx, by, bz = (_c(a, np.int32) for a in (bx, by, bz))
        bv = _c(bv, np.int8)
        assert len(bx) == len(by) == len(bz) == len(bv)
        self._ck(lib().pf_set_beams(self._h, _p(bx), _p(by), _p(bz), _p(bv), C.c_int64(len(bx))),
                 "pf_set_beams")
        self.M = len(bx)

    def measure(self, add_points):
        self._ck(lib().pf_measure(self._h, int(add_points)), "pf_measure")

    def get_scores(self):
        s = np.empty(self.P, np.int32); c = np.empty(self.P, np.int32)
        self._ck(lib().pf_get_scores(self._h, _p(s), _p(c)), "pf_get_scores")
        return s, c

    def scores_device(self):
        return lib().pf_scores_device(self._h)

    def poses_device(self):
        return lib().pf_poses_device(self._h)

    def _device_tensor(self, ptr, shape, typestr):
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False),
                                      "version": 2}
        return torch.as_tensor(v, device=f"cuda:{self.cfg.device}")

    def scores_tensor(self):
        """torch view (no copy) of the local scores, int32 [n_particles, 2], for the all-gather."""
        return self._device_tensor(self.scores_device(), (self.P, 2), "<i4")

    def poses_tensor(self):
        """torch view (no copy) of the local pose block, float64 [9, n_particles]."""
        return self._device_tensor(self.poses_device(), (9, self.P), "<f8")

    def voxel_indices(self, i):
        out = np.empty(self.M, np.int32)
        self._ck(lib().pf_voxel_indices(self._h, int(i), _p(out)), "pf_voxel_indices")
        return out

    # --- weights / resample ---
    def weights(self, scores_global_dev=None):
        self._ck(lib().pf_weights(self._h, C.c_void_p(scores_global_dev)), "pf_weights")

    def resample(self, u=None, poses_global_dev=None, want_slots=True):
        if u is not None:
            u = _c(u, np.float64)
            assert len(u) == self.Pg
        slots = np.empty(self.Pg, np.int32) if want_slots else None
        best = C.c_int32(-1); nohit = C.c_int64(0); amb = C.c_int64(0)
        self._ck(lib().pf_resample(self._h, _p(u), C.c_void_p(poses_global_dev), _p(slots),
                                   C.byref(best), C.byref(nohit), C.byref(amb)), "pf_resample")
        return slots, best.value, nohit.value, amb.value

    def resample_async(self, u=None, poses_global_dev=None, want_slots=False):
        """Enqueue only; the results (and the slots, when asked for) come with resample_result()."""
        if u is not None:
            u = _c(u, np.float64)
            assert len(u) == self.Pg
        slots = np.empty(self.Pg, np.int32) if want_slots else None
        self._ck(lib().pf_resample_async(self._h, _p(u), C.c_void_p(poses_global_dev), _p(slots)),
                 "pf_resample_async")
        self._pending_slots = getattr(self, "_pending_slots", [])
        self._pending_slots.append(slots)

    def resample_result(self):
        best = C.c_int32(-1); nohit = C.c_int64(0); amb = C.c_int64(0)
        self._ck(lib().pf_resample_result(self._h, C.byref(best), C.byref(nohit), C.byref(amb)),
                 "pf_resample_result")
        slots = self._pending_slots.pop(0) if getattr(self, "_pending_slots", None) else None
        return slots, best.value, nohit.value, amb.value

    # --- map update ---
    def set_scan(self, points, ray_idx):
        points = _c(points, np.float64).reshape(-1, 3)
        ray_idx = _c(ray_idx, np.int32)
        self._ck(lib().pf_set_scan(self._h, _p(points), C.c_int64(len(points)), _p(ray_idx),
                                   C.c_int64(len(ray_idx))), "pf_set_scan")

    def update_map(self, target, params, pos, zyx):
        pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
        self._ck(lib().pf_update_map(self._h, int(target), C.byref(params), _p(pos), _p(zyx)),
                 "pf_update_map")

    def scan_vote(self, points, pos, zyx, n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5, length=0.25):
        points = _c(points, np.float64).reshape(-1, 3)
        pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
        sums = np.empty(n_yaw * n_t ** 3, np.int32)
        best = C.c_int64(-1)
        self._ck(lib().pf_scan_vote(self._h, _p(points), C.c_int64(len(points)), _p(pos), _p(zyx),
                                    n_yaw, yaw_k0, yaw_kstep, n_t, C.c_double(length), _p(sums),
                                    C.byref(best)), "pf_scan_vote")
        return sums, best.value

    # --- keyframe ring of the localisation grid (S3P:305-312, 419-529) ---
    def ring_init(self, **kw):
        p = RingParams()
        lib().pf_ring_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self._ck(lib().pf_ring_init(self._h, C.byref(p)), "pf_ring_init")
        self._ring = p

    def ring_paint(self):
        self._ck(lib().pf_ring_paint(self._h), "pf_ring_paint")

    def ring_update(self, cloud, pos_scan, zyx):
        """Returns (flags, pos_scan after the call)."""
        cloud = _c(cloud, np.float64).reshape(-1, 3)
        pos = np.array(pos_scan, np.float64); zyx = _c(zyx, np.float64)
        flags = C.c_int32(0)
        self._ck(lib().pf_ring_update(self._h, _p(cloud), C.c_int64(len(cloud)), _p(pos), _p(zyx),
                                      C.byref(flags)), "pf_ring_update")
        return flags.value, pos

    def ring_state(self, with_idx=False):
        n, cap = self._ring.n_slots, self._ring.slot_capacity
        counts = np.empty(n, np.int32); lives = np.empty(n, np.int32)
        pu = np.empty(3, np.float64); eu = np.empty(3, np.float64)
        idx = np.empty(n * cap, np.int32) if with_idx else None
        self._ck(lib().pf_ring_state(self._h, _p(counts), _p(lives), _p(pu), _p(eu), _p(idx)), "pf_ring_state")
        return counts, lives, pu, eu, (idx.reshape(n, cap) if with_idx else None)

    def depth_to_cloud(self, image, width, height, pool=4, focal=385.0 / 4.0):
        """image: the uint16 little-endian depth bytes (height*width*2). Returns float64 [n][3]."""
        image = _c(np.frombuffer(image, np.uint8) if isinstance(image, (bytes, bytearray)) else image, np.uint8)
        if image.size != width * height * 2:
            raise PfError("image must hold width*height*2 bytes")
        cap = (width // pool) * (height // pool) if pool > 0 else 0
        out = np.empty((max(cap, 1), 3), np.float64)
        n = C.c_int64(0)
        self._ck(lib().pf_depth_to_cloud(self._h, _p(image), C.c_int32(width), C.c_int32(height), C.c_int32(pool),
                                         C.c_double(focal), _p(out), C.c_int64(cap), C.byref(n)),
                 "pf_depth_to_cloud")
        return out[:n.value].copy()

    # --- the same stages with the cloud kept in HBM (S3P:126): cloud_ptr is the caller's device buffer (an int) ---
    def depth_to_cloud_device(self, image, width, height, cloud_ptr, capacity, pool=4, focal=385.0 / 4.0):
        """Returns the number of rows written to the device buffer."""
        image = _c(np.frombuffer(image, np.uint8) if isinstance(image, (bytes, bytearray)) else image, np.uint8)
        if image.size != width * height * 2:
            raise PfError("image must hold width*height*2 bytes")
        n = C.c_int64(0)
        self._ck(lib().pf_depth_to_cloud_device(self._h, _p(image), C.c_int32(width), C.c_int32(height), C.c_int32(pool),
                                                C.c_double(focal), C.c_void_p(cloud_ptr), C.c_int64(capacity), C.byref(n)),
                 "pf_depth_to_cloud_device")
        return n.value

    def scan_vote_device(self, cloud_ptr, n_cloud, pick, pos, zyx, n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5, length=0.25):
        pick = _c(pick, np.int32)
        pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
        sums = np.empty(n_yaw * n_t ** 3, np.int32)
        best = C.c_int64(-1)
        self._ck(lib().pf_scan_vote_device(self._h, C.c_void_p(cloud_ptr), C.c_int64(n_cloud), _p(pick),
                                           C.c_int64(len(pick)), _p(pos), _p(zyx), n_yaw, yaw_k0, yaw_kstep, n_t,
                                           C.c_double(length), _p(sums), C.byref(best)), "pf_scan_vote_device")
        return sums, best.value

    def ring_update_device(self, cloud_ptr, n_points, pos_scan, zyx):
        """Returns (flags, pos_scan after the call)."""
        pos = np.array(pos_scan, np.float64); zyx = _c(zyx, np.float64)
        flags = C.c_int32(0)
        self._ck(lib().pf_ring_update_device(self._h, C.c_void_p(cloud_ptr), C.c_int64(n_points), _p(pos), _p(zyx),
                                             C.byref(flags)), "pf_ring_update_device")
        return flags.value, pos

    def set_scan_device(self, cloud_ptr, n_points, ray_idx):
        ray_idx = _c(ray_idx, np.int32)
        self._ck(lib().pf_set_scan_device(self._h, C.c_void_p(cloud_ptr), C.c_int64(n_points), _p(ray_idx),
                                          C.c_int64(len(ray_idx))), "pf_set_scan_device")

    def read_occupied(self, particle=0, threshold=80, stride=4, capacity=None):
        """Returns (float32 [n][3], total qualifying cells)."""
        cap = (self.elements + stride - 1) // stride if capacity is None and stride >= 1 else int(capacity or 0)
        out = np.empty((max(cap, 1), 3), np.float32)
        n = C.c_int64(0); total = C.c_int64(0)
        self._ck(lib().pf_read_occupied(self._h, C.c_int32(particle), C.c_int32(threshold), C.c_int32(stride),
                                        _p(out), C.c_int64(cap), C.byref(n), C.byref(total)), "pf_read_occupied")
        return out[:n.value].copy(), total.value

    def sequential_cdf(self, w, how=1):
        """S3P:763-767 on any weight vector: float64 [n+1]. how=0 one-lane chain, how=1 workgroup scan, how=2 many workgroups."""
        w = _c(w, np.float64).reshape(-1)
        out = np.empty(w.size + 1, np.float64)
        self._ck(lib().pf_sequential_cdf(self._h, _p(w), C.c_int64(w.size), C.c_int32(how), _p(out)),
                 "pf_sequential_cdf")
        return out

    def write_local_map(self, src):
        src = _c(src, np.int8).reshape(-1)
        assert src.size == self.local_elements
        self._ck(lib().pf_write_local_map(self._h, _p(src)), "pf_write_local_map")

    def read_local_map(self):
        dst = np.empty(self.local_elements, np.int8)
        self._ck(lib().pf_read_local_map(self._h, _p(dst)), "pf_read_local_map")
        return dst

    def reset_local_map(self):
        self._ck(lib().pf_reset_local_map(self._h), "pf_reset_local_map")

    # --- plumbing ---
    def stream(self):
        return lib().pf_stream(self._h)

    def set_stream(self, hip_stream):
        """Work on the caller's stream (an int/pointer, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(lib().pf_set_stream(self._h, C.c_void_p(hip_stream)), "pf_set_stream")
        self.external_stream = bool(hip_stream)

    def synchronize(self):
        self._ck(lib().pf_synchronize(self._h), "pf_synchronize")

    def profile(self, on, only=None):
        """on: bool; only: iterable of PF_K_* classes to time (default every class)."""
        v = int(bool(on))
        if on and only is not None:
            v = 2 * sum(1 << int(k) for k in only)
        self._ck(lib().pf_profile(self._h, v), "pf_profile")

    def kernel_time(self, kid):
        ms = C.c_double(0); n = C.c_int64(0)
        self._ck(lib().pf_kernel_time(self._h, int(kid), C.byref(ms), C.byref(n)), "pf_kernel_time")
        return ms.value, n.value


# ---- host-driver helpers (no GPU) ----------------------------------------------------------

def snapshot_beams(local, origin):
    local = _c(local, np.int8)
    lgx, lgy, lgz = local.shape
    cap = int(np.count_nonzero(local))
    bx = np.empty(cap, np.int32); by = np.empty(cap, np.int32); bz = np.empty(cap, np.int32)
    bv = np.empty(cap, np.int8)
    n = lib().pfh_snapshot_beams(_p(local), lgx, lgy, lgz, int(origin[0]), int(origin[1]),
                                 int(origin[2]), _p(bx), _p(by), _p(bz), _p(bv), C.c_int64(cap))
    assert n == cap, (n, cap)
    return bx, by, bz, bv


def host_predict(x, y, z, yaw, dl, dyaw, noise):
    noise = _c(noise, np.float32)
    for a in (x, y, z, yaw):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    lib().pfh_predict(C.c_int64(len(x)), _p(x), _p(y), _p(z), _p(yaw), C.c_double(dl[0]),
                      C.c_double(dl[1]), C.c_double(dl[2]), C.c_double(dyaw), _p(noise))


def draw_noise(seed, n, sigma_pos=0.03, sigma_rot=0.02):
    out = np.empty((n, 4), np.float32)
    lib().pfh_draw_noise(C.c_uint32(seed), C.c_int64(n), C.c_float(sigma_pos), C.c_float(sigma_rot),
                         _p(out))
    return out


def draw_uniform(seed, n):
    out = np.empty(n, np.float64)
    lib().pfh_draw_uniform(C.c_uint32(seed), C.c_int64(n), _p(out))
    return out


def add_points_policy(pose, last, threshold=0.4):
    pose = _c(pose, np.float64)
    assert last.dtype == np.float64 and last.flags.c_contiguous and len(last) == 4
    return int(lib().pfh_add_points_policy(_p(pose), _p(last), C.c_double(threshold)))


def vote_hypothesis(index, pos, zyx, resolution=0.05, n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5,
                    length=0.25):
    """(Tx, Ty, Tz, rot_z) of lattice index `index` (S3P:328-348); None for -1 / out of range."""
    pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
    out = np.empty(4, np.float64)
    rc = lib().pfh_vote_hypothesis(C.c_int64(int(index)), _p(pos), _p(zyx), C.c_double(resolution),
                                   n_yaw, yaw_k0, yaw_kstep, n_t, C.c_double(length), _p(out))
    return out if rc == 0 else None


def draw_indices(seed, n, hi):
    """n draws of uniform_int_distribution<>(0, hi) (inclusive) from mt19937(seed): S3P:315-319, 666-669."""
    idx = np.empty(n, np.int32)
    lib().pfh_draw_indices(C.c_uint32(seed), C.c_int64(n), C.c_int32(hi), _p(idx))
    return idx


def vote_epilogue(pos_scan, yaw_scan, loc, gyr_yaw, gyr_yaw_old):
    """S3P:379-404. Returns (pos_scan, yaw_scan, gyr_yaw_old, pos_diff[3], yaw_diff_scan, yaw_diff)."""
    pos = np.array(pos_scan, np.float64); loc = _c(loc, np.float64)
    yaw = C.c_double(yaw_scan); old = C.c_double(gyr_yaw_old)
    pd = np.empty(3, np.float64); ds = C.c_double(0); d = C.c_double(0)
    lib().pfh_vote_epilogue(_p(pos), C.byref(yaw), _p(loc), C.c_double(gyr_yaw), C.byref(old), _p(pd),
                            C.byref(ds), C.byref(d))
    return pos, yaw.value, old.value, pd, ds.value, d.value


def update_realtime_pos(pos, yaw, pos_diff_scan, yaw_diff, feedback):
    """S3P:712-728. Returns (pos[3], yaw); `feedback` (float64 [4]) is consumed in place."""
    pos = np.array(pos, np.float64); pd = _c(pos_diff_scan, np.float64)
    assert feedback.dtype == np.float64 and feedback.flags.c_contiguous and len(feedback) == 4
    y = C.c_double(yaw)
    lib().pfh_update_realtime_pos(_p(pos), C.byref(y), _p(pd), C.c_double(yaw_diff), _p(feedback))
    return pos, y.value


def pose_message(pos, yaw):
    """S3P:207-222: (position[3], orientation wxyz[4]) of the published PoseStamped."""
    pos = _c(pos, np.float64)
    p = np.empty(3, np.float64); q = np.empty(4, np.float64)
    lib().pfh_pose_message(_p(pos), C.c_double(yaw), _p(p), _p(q))
    return p, q


def migration_plan(slot_anc, n_local, rank):
    """pfh_migration_plan: (total transfers, sends [(src_local, peer, seq)], recvs [(dst_local, peer, seq)])."""
    sa = _c(slot_anc, np.int32)
    cap = len(sa)
    ss = np.empty(cap, np.int32); sp = np.empty(cap, np.int32); sq = np.empty(cap, np.int64)
    rd = np.empty(cap, np.int32); rp = np.empty(cap, np.int32); rq = np.empty(cap, np.int64)
    ns = C.c_int64(0); nr = C.c_int64(0)
    total = lib().pfh_migration_plan(_p(sa), C.c_int64(cap), C.c_int64(n_local), C.c_int32(rank), _p(ss), _p(sp),
                                     _p(sq), C.byref(ns), _p(rd), _p(rp), _p(rq), C.byref(nr), C.c_int64(cap))
    return (total, list(zip(ss[:ns.value].tolist(), sp[:ns.value].tolist(), sq[:ns.value].tolist())),
            list(zip(rd[:nr.value].tolist(), rp[:nr.value].tolist(), rq[:nr.value].tolist())))


def scan_vote_static(device, stream, grid_ptr, dims, resolution, points_ptr, n_points, pos, zyx, sums_ptr,
                     n_yaw=5, yaw_k0=-6, yaw_kstep=3, n_t=5, length=0.25):
    """pf_scan_vote_static: device pointers in and out (ints), nothing allocated. Raises PfError."""
    pos = _c(pos, np.float64); zyx = _c(zyx, np.float64)
    rc = lib().pf_scan_vote_static(C.c_int32(device), C.c_void_p(stream), C.c_void_p(grid_ptr), C.c_int32(dims[0]),
                                   C.c_int32(dims[1]), C.c_int32(dims[2]), C.c_double(resolution),
                                   C.c_void_p(points_ptr), C.c_int64(n_points), _p(pos), _p(zyx), C.c_int32(n_yaw),
                                   C.c_int32(yaw_k0), C.c_int32(yaw_kstep), C.c_int32(n_t), C.c_double(length),
                                   C.c_void_p(sums_ptr))
    if rc != PF_OK:
        raise PfError(f"pf_scan_vote_static -> {rc}: {lib().pf_last_error(None).decode()}")


def scan_vote_static_warmup(device):
    """pf_scan_vote_static_warmup: the one-time set-up of pf_scan_vote_static on a device (outside any capture)."""
    rc = lib().pf_scan_vote_static_warmup(C.c_int32(device))
    if rc != PF_OK:
        raise PfError(f"pf_scan_vote_static_warmup -> {rc}: {lib().pf_last_error(None).decode()}")


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().pf_comm_unique_id(buf, C.c_int64(128))
    if rc != 0:
        raise RuntimeError(f"pf_comm_unique_id failed ({rc}): {lib().pf_last_error(None).decode()}")
    return buf.raw


def deal_groups(group_ends, n_chunks, trips):
    """pfh_deal_groups: (perm, table [n_chunks][9]) of K1's chunk-major group list."""
    ge = _c(group_ends, np.int32); tr = _c(trips, np.int32)
    assert ge.size == 8 and tr.size == 8
    perm = np.empty(int(ge[7]), np.int32); table = np.empty((max(int(n_chunks), 1), 9), np.int32)
    lib().pfh_deal_groups(_p(ge), C.c_int32(n_chunks), _p(tr), _p(perm), _p(table))
    return perm, table


def permute(slot_anc, records):
    assert records.flags.c_contiguous
    slot_anc = _c(slot_anc, np.int32)
    lib().pfh_permute(C.c_int64(len(slot_anc)), _p(slot_anc), _p(records),
                      C.c_size_t(records.strides[0]))
